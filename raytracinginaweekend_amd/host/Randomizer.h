// Randomizer.h -- host draws for scene construction with the reference's names and mappings
// (RayTracer/RayTracer/Randomizer.h:9-58).  The reference constructs a std::random_device per draw;
// here the source is one process-wide Philox4x32-10 stream (Seed() makes scenes reproducible, the
// default seed comes from std::random_device once).  Per-path draws happen on the device.
#pragma once
#include "Vec3.h"

class Randomizer
{
public:
	static void Seed(UINT64 seed);
	static UINT32 Raw();                                     // one `rd()`
	static float RandomMinMax(float min, float max);          // [min, max)
	static float RandomMinMax2(float min, float max);         // [min, max]
	static float RandomUNorm() { return RandomMinMax(0.0f, 1.0f); }
	static float RandomUNorm2() { return RandomMinMax2(0.0f, 1.0f); }
	static Vec3 RomdomInUnitSphere();
	static Vec3 RandomInUnitDisk();
};
