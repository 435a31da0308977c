// SimpleTexture2D.h -- ITexture2D, the single-colour texture and the TGA image texture
// (reference SimpleTexture2D.h:13-60, SimpleTexture2D.cpp:109-147).  On the render path Sample() runs on the
// device (rt_device.h tex_sample); the texels are handed over by FlatScene.
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

// Decodes a TGA file image (types 1, 2, 3 and their RLE forms 9, 10, 11; 8-bit grey, 24/32-bit colour, 24/32-bit
// palettes; any origin corner) into R,G,B,A bytes, row 0 = top -- what the reference gets from
// tgaRead(buffer, TGA_READER_ABGR) (tga_reader.cpp, npedotnet/TGAReader).  Throws std::runtime_error on a bad file.
void DecodeTGA(const std::vector<UINT8> &file, UINT32 &width, UINT32 &height, std::vector<UINT8> &rgba);

class SimpleTexture2D_TGAImage : public SimpleTexture2D
{
public:
	explicit SimpleTexture2D_TGAImage(const char *filePath);                       // SimpleTexture2D.cpp:109-118
	SimpleTexture2D_TGAImage(UINT32 width, UINT32 height, const UINT8 *rgba);      // from decoded texels
	Vec3 Sample(float u, float v) const override;                                  // SimpleTexture2D.cpp:130-147
	std::vector<UINT8> m_pixelData;
};
