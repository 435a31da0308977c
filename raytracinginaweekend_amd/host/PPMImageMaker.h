// PPMImageMaker.h -- ASCII P3 writer (reference PPMImageMaker.cpp:9-48)
#pragma once
#include "stdafx.h"

class PPMImageMaker
{
public:
	static void OutputRGBA8ToFile(const char *fileName, UINT32 imageWidth, UINT32 imageHeight, const UINT8 *rgba8PixelData, UINT64 dataSizeInByte = 0);
};
