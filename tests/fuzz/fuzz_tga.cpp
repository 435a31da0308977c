#include "stdafx.h"
#include "SimpleTexture2D.h"
#include <random>
#include <fstream>
int main(int argc, char **argv)
{
	std::vector<std::vector<UINT8>> seeds;
	for (int i = 1; i < argc; ++i) { std::ifstream f(argv[i], std::ios::binary); seeds.emplace_back((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
	std::mt19937 rng(1234);
	size_t ok = 0, bad = 0;
	const int iters = 20000;
	for (int it = 0; it < iters; ++it) {
		std::vector<UINT8> f = seeds[rng() % seeds.size()];
		int nmut = 1 + rng() % 4;
		for (int m = 0; m < nmut; ++m) {
			switch (rng() % 4) {
			case 0: if (!f.empty()) f[rng() % std::min<size_t>(f.size(), 40)] = (UINT8)rng(); break;      // header / palette bytes
			case 1: if (!f.empty()) f[rng() % f.size()] = (UINT8)rng(); break;
			case 2: f.resize(rng() % (f.size() + 1)); break;
			case 3: f.insert(f.end(), rng() % 64, (UINT8)rng()); break;
			}
		}
		if (f.size() >= 16) { f[13] &= 0x01; f[15] &= 0x01; }     // keep images under 512x512 so the run stays fast
		UINT32 w = 0, h = 0; std::vector<UINT8> rgba;
		try { DecodeTGA(f, w, h, rgba); if (rgba.size() != (size_t)w * h * 4) { printf("size mismatch\n"); return 2; } ++ok; }
		catch (const std::runtime_error &) { ++bad; }
	}
	printf("decoded %zu, rejected %zu\n", ok, bad);
	return 0;
}
