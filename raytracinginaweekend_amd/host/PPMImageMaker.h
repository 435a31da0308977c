// PPMImageMaker.h -- ASCII P3 writer (reference PPMImageMaker.cpp:9-48)
#pragma once
#include "stdafx.h"

class PPMImageMaker
{
public:
	static void OutputRGBA8ToFile(const char *fileName, UINT32 imageWidth, UINT32 imageHeight, const UINT8 *rgba8PixelData, UINT64 dataSizeInByte = 0);
	// not in the reference: binary P6 of the same pixels (12 bytes of text per pixel become 3 bytes)
	static void OutputRGBA8ToBinaryFile(const char *fileName, UINT32 imageWidth, UINT32 imageHeight, const UINT8 *rgba8PixelData, UINT64 dataSizeInByte = 0);
};
