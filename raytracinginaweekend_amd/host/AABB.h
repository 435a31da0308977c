// AABB.h -- RayTracer/RayTracer/AABB.h:7-34.  The slab test (AABB.cpp:5-21) runs on the device
// (rt_device.h aabb_hit); the host keeps the box type and the union used to build trees.
#pragma once
#include "Vec3.h"

class AABB
{
public:
	AABB() = default;
	AABB(const Vec3 &mn, const Vec3 &mx) : m_min(mn), m_max(mx) {}
	Vec3 m_min, m_max;
};

inline AABB CombineAABB(const AABB &a, const AABB &b)
{
	Vec3 mn, mx;
	for (int i = 0; i < 3; ++i) { mn.set(i, a.m_min[i] < b.m_min[i] ? a.m_min[i] : b.m_min[i]); mx.set(i, a.m_max[i] > b.m_max[i] ? a.m_max[i] : b.m_max[i]); }
	return AABB(mn, mx);
}
