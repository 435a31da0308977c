// SimpleObject.h -- Object, the sphere / rect / cube factories and the BVH node of the reference
// (RayTracer/RayTracer/SimpleObject.h:27-101), minus the D3D12 halves.  `Mesh*` arguments are accepted
// and ignored: meshes only feed the rasterizer preview.
#pragma once
#include "Vec3.h"
#include "AABB.h"
#include "FlatScene.h"

class Mesh;
class IHitable;
class IMaterial;
class World;

class Object
{
public:
	virtual ~Object();
	World     *m_world{ nullptr };
	Vec3       m_translation;
	Vec3       m_scaling;
	Vec3       m_rotation;
	Mesh      *m_mesh{ nullptr };
	IHitable  *m_hitable{ nullptr };
	IMaterial *m_material{ nullptr };

	virtual AABB BoundingBox() const;
	// flattens the subtree rooted here; returns the child code of include/mirt.h (node index or ~object)
	virtual int32_t Flatten(FlatScene &out) const;
};

class SimpleObjectSphere : public Object
{
public:
	SimpleObjectSphere(const Vec3 &center, float radius, Mesh *mesh, IMaterial *material, World *world);
};

enum SimpleObjectRectAlignAxes
{
	XY_RECT,
	XZ_RECT,
	ZY_RECT,
};

class SimpleObjectRect : public Object
{
public:
	SimpleObjectRect(SimpleObjectRectAlignAxes axes, const Vec3 &center, const Vec3 &rotation, float width, float height, BOOL reverseFace,
	                 Mesh *mesh, IMaterial *material, World *world);
	SimpleObjectRectAlignAxes m_alignAxes;
	BOOL                      m_reverseFace;
};

class SimpleObjectCube : public Object
{
public:
	SimpleObjectCube(const Vec3 &center, const Vec3 &rotation, const Vec3 &size, Mesh *mesh, IMaterial *material, World *world);
	~SimpleObjectCube() override;
	IHitable *m_faceList[6];
};

class SimpleObjectBVHNode : public Object
{
public:
	explicit SimpleObjectBVHNode(std::vector<Object *> objects);
	~SimpleObjectBVHNode() override;
	AABB BoundingBox() const override { return m_bindingBox; }
	int32_t Flatten(FlatScene &out) const override;

	AABB    m_bindingBox;
	Object *leftChild{ nullptr };
	Object *rightChild{ nullptr };
};
