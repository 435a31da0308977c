// OutputImage.h -- the RGBA8 image sink (reference OutputImage.h:9-42) without the D3D12 upload.
#pragma once
#include "stdafx.h"

struct Vec3;

class OutputImage
{
public:
	OutputImage(UINT32 width, UINT32 height, const char *name);
	~OutputImage();

	void RenderAsRainbow();
	void RenderAsRed();
	void Render(const Vec3 *pixels, UINT32 pixelCount = 0);
	void Output();                                   // <name>.ppm through PPMImageMaker
	void OutputBinary();                             // not in the reference: <name>.ppm as binary P6

	UINT32      m_width{ 0 };
	UINT32      m_height{ 0 };
	float       m_aspectRatio{ 0.0f };
	UINT64      m_dataSizeInByte{ 0 };
	UINT32      m_pixelSizeInByte{ 4 };
	UINT8      *m_data{ nullptr };
	std::string m_name{};
	BOOL        m_isDirty{ FALSE };
	std::string m_outputDir{ "." };                  // the reference hard-codes ..\Assets
};
