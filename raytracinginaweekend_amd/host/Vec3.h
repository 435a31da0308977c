// Vec3.h -- host-side Vec3 with the interface of the reference's Vec3 (RayTracer/RayTracer/Vec3.h:5-108)
// on three plain floats.  Operation order follows the DirectXMath SSE2 path the reference runs on:
// dot = (x*x' + y*y') + z*z', v/f = v*(1/f), -v = 0 - v, normalize = v / sqrt(dot).
#pragma once
#include "stdafx.h"

struct Vec3
{
	float e[3];

	Vec3() = default;
	Vec3(float e0, float e1, float e2) { e[0] = e0; e[1] = e1; e[2] = e2; }

	float x() const { return e[0]; }
	float y() const { return e[1]; }
	float z() const { return e[2]; }
	float r() const { return e[0]; }
	float g() const { return e[1]; }
	float b() const { return e[2]; }

	const Vec3 &operator+() const { return *this; }
	Vec3 operator-() const { return Vec3(0.0f - e[0], 0.0f - e[1], 0.0f - e[2]); }
	float operator[](size_t i) const { return e[i]; }
	void set(size_t i, float f) { e[i] = f; }

	Vec3 &operator+=(const Vec3 &v) { for (int i = 0; i < 3; ++i) e[i] += v.e[i]; return *this; }
	Vec3 &operator-=(const Vec3 &v) { for (int i = 0; i < 3; ++i) e[i] -= v.e[i]; return *this; }
	Vec3 &operator*=(const Vec3 &v) { for (int i = 0; i < 3; ++i) e[i] *= v.e[i]; return *this; }
	Vec3 &operator/=(const Vec3 &v) { for (int i = 0; i < 3; ++i) e[i] /= v.e[i]; return *this; }
	Vec3 &operator*=(float f) { for (int i = 0; i < 3; ++i) e[i] *= f; return *this; }
	Vec3 &operator/=(float f) { float k = 1.f / f; for (int i = 0; i < 3; ++i) e[i] *= k; return *this; }

	float squared_length() const { return (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]; }
	float length() const { return sqrtf(squared_length()); }
	void normalize()
	{
		float l2 = squared_length(), l = sqrtf(l2);
		if (l == 0.0f) { zero(); return; }
		if (std::isinf(l2)) { e[0] = e[1] = e[2] = NAN; return; }
		for (int i = 0; i < 3; ++i) e[i] /= l;
	}
	void zero() { e[0] = e[1] = e[2] = 0.0f; }
	void clamp(const Vec3 &lo, const Vec3 &hi)
	{
		for (int i = 0; i < 3; ++i) { float m = (lo.e[i] > e[i]) ? lo.e[i] : e[i]; e[i] = (hi.e[i] < m) ? hi.e[i] : m; }
	}
};

inline std::ostream &operator<<(std::ostream &os, const Vec3 &v) { return os << "[" << v.x() << "," << v.y() << "," << v.z() << "]"; }
inline std::istream &operator>>(std::istream &is, Vec3 &v) { return is >> v.e[0] >> v.e[1] >> v.e[2]; }
inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline Vec3 operator*(const Vec3 &a, const Vec3 &b) { return Vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline Vec3 operator/(const Vec3 &a, const Vec3 &b) { return Vec3(a.e[0] / b.e[0], a.e[1] / b.e[1], a.e[2] / b.e[2]); }
inline Vec3 operator*(float f, const Vec3 &v) { return Vec3(v.e[0] * f, v.e[1] * f, v.e[2] * f); }
inline Vec3 operator*(const Vec3 &v, float f) { return Vec3(v.e[0] * f, v.e[1] * f, v.e[2] * f); }
inline Vec3 operator/(const Vec3 &v, float f) { float k = 1.f / f; return Vec3(v.e[0] * k, v.e[1] * k, v.e[2] * k); }
inline float dot(const Vec3 &a, const Vec3 &b) { return (a.e[0] * b.e[0] + a.e[1] * b.e[1]) + a.e[2] * b.e[2]; }
inline Vec3 cross(const Vec3 &a, const Vec3 &b)
{
	return Vec3(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
inline Vec3 normalize(const Vec3 &v) { Vec3 r = v; r.normalize(); return r; }
inline Vec3 zero() { return Vec3(0.0f, 0.0f, 0.0f); }

// Ray (RayTracer/RayTracer/Ray.h:5-18): origin + t * direction; the direction is not normalised.
struct Ray
{
	Vec3 m_org, m_dir;
	Ray() = default;
	Ray(const Vec3 &org, const Vec3 &dir) : m_org(org), m_dir(dir) {}
	Vec3 PointAt(float t) const { return m_org + t * m_dir; }
};
