// World.h -- reference World.h:9-44 without the D3D12 members.
#pragma once
#include "stdafx.h"
#include "FlatScene.h"
#include "Ray.h"

struct mirt_ctx;

class Resources;
class SimpleCamera;
class SimpleObjectBVHNode;
class LightSources;

enum WorldID
{
	WORLD_ID_RANDOM_SPHERES = 0,
	WORLD_ID_CORNELL_BOX,
	// additions of this build (BASELINE.json configs 1 and the `#if 0` variant of reference World.cpp:83-89)
	WORLD_ID_FOUR_SPHERES,
	WORLD_ID_CORNELL_TWO_BOXES,
	// the two image-textured materials the reference defines but no shipped world uses (Resouces.cpp:166-176):
	// a soil-textured floor and back wall, a checker-plate metal ball, a white light, sky ambient
	WORLD_ID_TEXTURED,
	// the hitable vocabulary beyond the factories' shapes (Hitables.h:20-106 compose freely): a sphere with its own centre inside
	// a rotation inside a translation, combos of spheres, nested combos under instances, rotations out of the factories' z,y,x order
	WORLD_ID_CHAINS,
};

class World
{
public:
	World() = default;
	~World() = default;

	void ConstructWorld(WorldID wid, SimpleCamera *camera);
	void DeconstructWorld();
	void OnUpdate(SimpleCamera *, float) {}

	SimpleObjectBVHNode *GetObjectBVHTree() const { return m_objectBVHTree; }
	LightSources        *GetLightSources() const { return m_lightSources; }
	size_t               GetObjectsCount() const { return m_objectsCount; }
	// for scene code that builds its own world: hand over the tree and the lights
	void Adopt(SimpleObjectBVHNode *tree, LightSources *lights, size_t objectCount) { m_objectBVHTree = tree; m_lightSources = lights; m_objectsCount = objectCount; ++m_generation; }
	// Bumped whenever the scene changes (ConstructWorld, DeconstructWorld, Adopt); HomemadeRayTracer re-flattens and
	// re-uploads when it differs from the generation it uploaded.  Scene code that edits objects in place calls Touch().
	UINT64 Generation() const { return m_generation; }
	void   Touch() { ++m_generation; if (m_queryCtx) { mirt_destroy(m_queryCtx); m_queryCtx = nullptr; } }

	void Flatten(FlatScene &out) const;

	// What SimpleCamera::AutoFocus asks of GetObjectBVHTree()->Hit (SimpleCamera.cpp:202-214): the closest hit of
	// one ray.  Hit() exists only on the device, so this is a one-ray mirt_world_hit_batch on a context the world
	// keeps for such queries (the tree is walked in the reference's order).  Throws without a GPU.
	BOOL ClosestHit(const Ray &r, float tMin, float tMax, Vec3 &position) const;
	void SetQueryDevice(int device) { m_queryDevice = device; }

private:
	Resources           *m_resources{ nullptr };
	SimpleObjectBVHNode *m_objectBVHTree{ nullptr };
	size_t               m_objectsCount{ 0 };
	std::vector<class Object *> m_objects;            // construction order (owned by the tree)
	LightSources        *m_lightSources{ nullptr };
	mutable mirt_ctx    *m_queryCtx{ nullptr };
	int                  m_queryDevice{ 0 };
	UINT64               m_generation{ 0 };
};
