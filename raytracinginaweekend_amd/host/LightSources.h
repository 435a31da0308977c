// LightSources.h -- the light list and the constant ambient colour (reference LightSources.h:10-41).
#pragma once
#include "Vec3.h"

class Object;
class World;

class LightSources
{
public:
	LightSources(World *world, std::vector<Object *> &objects, const Vec3 &ambientLight);
	UINT32      GetLightSourceCount() const { return m_lightSourceCountX; }
	const Vec3 &GetAmbientLight() const { return m_ambientLight; }

private:
	World                *m_world{ nullptr };
	std::vector<Object *> m_lightSources;
	UINT32                m_lightSourceCount{ 0 };
	UINT32                m_lightSourceCountX{ 0 };
	Vec3                  m_ambientLight;
};
