// Ray.h -- kept so that `#include "Ray.h"` of scene code written against the reference resolves; Ray lives beside Vec3.
#pragma once
#include "Vec3.h"
