// SimpleTexture2D.h -- ITexture2D and the single-colour texture (reference SimpleTexture2D.h:13-41).
// Image (TGA) textures are not used by either shipped world and are not on the device path yet.
#pragma once
#include "Vec3.h"

class ITexture2D
{
public:
	virtual ~ITexture2D() = default;
	virtual Vec3 Sample(float u, float v) const = 0;
	virtual bool IsSingleColor() const { return false; }
};

class SimpleTexture2D : public ITexture2D
{
public:
	UINT32 m_width{ 0 }, m_height{ 0 };
};

class SimpleTexture2D_SingleColor : public SimpleTexture2D
{
public:
	explicit SimpleTexture2D_SingleColor(const Vec3 &col) : m_color(col) { m_width = m_height = 4; }
	Vec3 Sample(float, float) const override { return m_color; }
	bool IsSingleColor() const override { return true; }
	Vec3 m_color;
};
