// stdafx.h -- the build's own prefix header for scene code written against the reference: the few
// Win32 spellings its class interfaces use (BOOL, UINT32, ...) plus the standard headers.  It is not
// a stand-in for <windows.h> or DirectXMath: nothing of the reference is compiled against it.
#pragma once
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

typedef int BOOL;
typedef uint8_t UINT8;
typedef uint16_t UINT16;
typedef uint32_t UINT32;
typedef uint32_t UINT;
typedef int32_t INT32;
typedef uint64_t UINT64;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#ifndef VK_SPACE
#define VK_SPACE 0x20
#define VK_ESCAPE 0x1B
#define VK_LEFT 0x25
#define VK_UP 0x26
#define VK_RIGHT 0x27
#define VK_DOWN 0x28
#endif
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
