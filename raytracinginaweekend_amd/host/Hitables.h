// Hitables.h -- the hitable vocabulary of the reference (RayTracer/RayTracer/Hitables.h:9-106) as
// scene-description classes.  Hit() itself runs on the device (rt_device.h prim_t / prim_record); on
// the host a hitable knows its bounding box and how to flatten itself through its instance chain.
#pragma once
#include "Vec3.h"
#include "AABB.h"
#include "FlatScene.h"

struct Ray;
class IMaterial;

struct HitRecord
{
	float      m_time;
	Vec3       m_position;
	Vec3       m_normal;
	float      m_u;
	float      m_v;
	IMaterial *m_hitMaterial;
};

class IHitable
{
public:
	IMaterial *m_material{ nullptr };

	virtual ~IHitable() = default;
	virtual void BindMaterial(IMaterial *m) { m_material = m; }
	virtual AABB BoundingBox() const = 0;
	// appends this hitable's primitives to `out` (inside the object the caller has begun)
	virtual void Flatten(FlatScene &out, const InstanceChain &chain) const = 0;
};

class SphereHitable : public IHitable
{
public:
	Vec3  m_center;
	float m_radius;

	SphereHitable(const Vec3 &center, float radius) : m_center(center), m_radius(radius) {}
	AABB BoundingBox() const override;
	void Flatten(FlatScene &out, const InstanceChain &chain) const override;
};

class AxisAlignedRectHitable : public IHitable
{
public:
	float  m_a0, m_a1, m_b0, m_b1, m_c;
	UINT32 m_aAxisIndex, m_bAxisIndex, m_cAxisIndex;
	BOOL   m_reverseFace;

	AxisAlignedRectHitable(UINT32 aAxisIndex, UINT32 bAxisIndex, float a0, float a1, float b0, float b1, float c, BOOL reverseFace);
	AABB BoundingBox() const override;
	void Flatten(FlatScene &out, const InstanceChain &chain) const override;
};

class HitableCombo : public IHitable
{
public:
	IHitable **m_hitableList{ nullptr };
	UINT32     m_hitableCount{ 0 };

	HitableCombo(IHitable **plist, UINT32 count) : m_hitableList(plist), m_hitableCount(count) {}
	void BindMaterial(IMaterial *m) override;
	AABB BoundingBox() const override;
	void Flatten(FlatScene &out, const InstanceChain &chain) const override;
};

class TransformedInstance : public IHitable
{
public:
	IHitable *m_hitable;

	explicit TransformedInstance(IHitable *hitable) : m_hitable(hitable) {}
	~TransformedInstance() override { delete m_hitable; }
	void BindMaterial(IMaterial *m) override { if (m_hitable) m_hitable->BindMaterial(m); }
};

class TranslatedInstance : public TransformedInstance
{
public:
	Vec3 m_offset;

	TranslatedInstance(IHitable *hitable, const Vec3 &displacement) : TransformedInstance(hitable), m_offset(displacement) {}
	AABB BoundingBox() const override;
	void Flatten(FlatScene &out, const InstanceChain &chain) const override;
};

class RotatedInstance : public TransformedInstance
{
public:
	float  m_sinTheta, m_cosTheta;
	AABB   m_boundingBox;
	UINT32 m_aAxisIndex, m_bAxisIndex, m_cAxisIndex;
	float  m_OP, m_oppositeOP;

	RotatedInstance(IHitable *hitable, float angle, UINT32 rotateAxis);
	AABB BoundingBox() const override { return m_boundingBox; }
	void Flatten(FlatScene &out, const InstanceChain &chain) const override;
};
