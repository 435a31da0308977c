// World.h -- reference World.h:9-44 without the D3D12 members.
#pragma once
#include "stdafx.h"
#include "FlatScene.h"

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
	void Adopt(SimpleObjectBVHNode *tree, LightSources *lights, size_t objectCount) { m_objectBVHTree = tree; m_lightSources = lights; m_objectsCount = objectCount; }

	void Flatten(FlatScene &out) const;

private:
	Resources           *m_resources{ nullptr };
	SimpleObjectBVHNode *m_objectBVHTree{ nullptr };
	size_t               m_objectsCount{ 0 };
	std::vector<class Object *> m_objects;            // construction order (owned by the tree)
	LightSources        *m_lightSources{ nullptr };
};
