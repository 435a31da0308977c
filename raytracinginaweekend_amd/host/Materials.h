// Materials.h -- the four materials of the reference (RayTracer/RayTracer/Materials.h:7-124) as
// parameter holders.  Scatter() runs on the device (rt_device.h scatter); the D3D12 half is dropped.
#pragma once
#include "Vec3.h"
#include "../../include/mirt.h"

enum MaterialID
{
	MID_DIFFUSE_LIGHT = 0,
	MID_LAMBERTIAN,
	MID_METAL,
	MID_DIELECTRIC,

	MID_COUNT
};

class ITexture2D;

struct FlatMaterial { uint32_t kind; float albedo[3]; float fuzz, ior; float emit[3]; const ITexture2D *image; /* non-null: an image texture supplies the albedo */ };

class IMaterial
{
public:
	virtual ~IMaterial() = default;
	virtual MaterialID GetID() const = 0;
	virtual FlatMaterial Flatten() const = 0;
};

class Lambertian : public IMaterial
{
public:
	const ITexture2D *m_albedo{ nullptr };
	explicit Lambertian(const ITexture2D *albedo) : m_albedo(albedo) {}
	MaterialID GetID() const override { return GetStaticID(); }
	static MaterialID GetStaticID() { return MID_LAMBERTIAN; }
	FlatMaterial Flatten() const override;
};

class Metal : public IMaterial
{
public:
	float m_fuzziness;
	ITexture2D *m_albedo{ nullptr };
	Metal(ITexture2D *albedo, float fuzziness);
	MaterialID GetID() const override { return GetStaticID(); }
	static MaterialID GetStaticID() { return MID_METAL; }
	FlatMaterial Flatten() const override;
};

class Dielectric : public IMaterial
{
public:
	float m_refractiveIndex;
	explicit Dielectric(float refractiveIndex) : m_refractiveIndex(refractiveIndex) {}
	MaterialID GetID() const override { return GetStaticID(); }
	static MaterialID GetStaticID() { return MID_DIELECTRIC; }
	FlatMaterial Flatten() const override;
};

class DiffuseLight : public IMaterial
{
public:
	Vec3 m_intensity;
	explicit DiffuseLight(const Vec3 intensity) : m_intensity(intensity) {}
	MaterialID GetID() const override { return GetStaticID(); }
	static MaterialID GetStaticID() { return MID_DIFFUSE_LIGHT; }
	FlatMaterial Flatten() const override;
};

// The slots of Resources' tables (RayTracer/RayTracer/Resouces.h:3-43): the enumerators are part of the scene API
// (`GetTheMaterial(MATERIAL_ID_DIELECTRIC)`), and their order is the order Resources::LoadMaterials fills the table in.
enum MeshUniqueID
{
	MESH_ID_HIGH_POLYGON_SPHERE = 0,
	MESH_ID_MEDIUM_POLYGON_SPHERE,
	MESH_ID_LOW_POLYGON_SPHERE,
	MESH_ID_QUAD,
	MESH_ID_CUBE,

	MESH_ID_TOTAL_COUNT
};

enum MaterialUniqueID
{
	MATERIAL_ID_RANDOM_LAMBERTIAN_START = 0,
	MATERIAL_ID_RANDOM_LAMBERTIAN_COUNT = 50,
	MATERIAL_ID_RANDOM_METAL_START = MATERIAL_ID_RANDOM_LAMBERTIAN_START + MATERIAL_ID_RANDOM_LAMBERTIAN_COUNT,
	MATERIAL_ID_RANDOM_METAL_COUNT = 50,
	MATERIAL_ID_LAMBERTIAN0 = MATERIAL_ID_RANDOM_METAL_START + MATERIAL_ID_RANDOM_METAL_COUNT,
	MATERIAL_ID_LAMBERTIAN1,
	MATERIAL_ID_LAMBERTIAN2,
	MATERIAL_ID_LAMBERTIAN3,
	MATERIAL_ID_LAMBERTIAN4,
	MATERIAL_ID_METAL,
	MATERIAL_ID_DIELECTRIC,
	MATERIAL_ID_IMAGE_BASED_GROUND_SOIL,
	MATERIAL_ID_IMAGE_BASED_METAL_CHECKER,
	MATERIAL_ID_LIGHTSOURCE_WHITE,
	MATERIAL_ID_LIGHTSOURCE_BRIGHT,
	MATERIAL_ID_LIGHTSOURCE_RED,
	MATERIAL_ID_LIGHTSOURCE_GREEN,
	MATERIAL_ID_LIGHTSOURCE_BLUE,
	MATERIAL_ID_CORNELL_BALL1,
	MATERIAL_ID_CORNELL_BALL2,

	MATERIAL_ID_TOTAL_COUNT
};

