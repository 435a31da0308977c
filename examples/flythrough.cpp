// flythrough.cpp -- the reference's viewer loop without the window (RayTracer.cpp:251-254 re-invokes the
// tracer from WM_PAINT while the camera's hot keys move it): a scripted, headless fly-through.
// Every frame holds a set of keys for 1/30 s, the camera moves and refocuses (SimpleCamera::OnUpdate,
// AutoFocus), a 1-SPP preview is traced and saved as binary PPM; the last view is then refined
// progressively (passes of `pass` samples up to `spp`), saving the image after every pass.
//
//   flythrough [world width height frames spp pass outdir [script]]
//     world: 0 random spheres, 1 Cornell box, 2 four spheres, 3 Cornell with two boxes
//     script: keys held, one group per frame, cycled: e.g. "W,W,WL,L,-" (W A S D move, L R U P = arrow
//             left/right/up/down, - = none); default "W,WL,L,D,-"
#include "stdafx.h"
#include "HomemadeRayTracer.h"
#include "InputListener.h"
#include "OutputImage.h"
#include "SimpleCamera.h"
#include "World.h"
#include <chrono>
#include <sstream>

static std::vector<std::vector<UINT8>> ParseScript(const std::string &script)
{
	std::vector<std::vector<UINT8>> frames;
	std::stringstream ss(script);
	std::string group;
	while (std::getline(ss, group, ',')) {
		std::vector<UINT8> keys;
		for (char c : group) {
			switch (c) {
			case 'W': case 'A': case 'S': case 'D': keys.push_back((UINT8)c); break;
			case 'L': keys.push_back(VK_LEFT); break;
			case 'R': keys.push_back(VK_RIGHT); break;
			case 'U': keys.push_back(VK_UP); break;
			case 'P': keys.push_back(VK_DOWN); break;
			case '-': break;
			default: throw std::invalid_argument(std::string("flythrough: unknown key '") + c + "' in script");
			}
		}
		frames.push_back(keys);
	}
	if (frames.empty()) frames.push_back({});
	return frames;
}

int main(int argc, char **argv)
{
	const int    worldId = argc > 1 ? atoi(argv[1]) : 1;
	const UINT32 width = argc > 3 ? atoi(argv[2]) : 512, height = argc > 3 ? atoi(argv[3]) : 512;
	const UINT32 frames = argc > 4 ? atoi(argv[4]) : 8, spp = argc > 5 ? atoi(argv[5]) : 256, pass = argc > 6 ? std::max(1, atoi(argv[6])) : 64;
	const std::string outDir = argc > 7 ? argv[7] : ".";
	const float dt = 1.0f / 30.0f;
	try {
		const auto script = ParseScript(argc > 8 ? argv[8] : "W,WL,L,D,-");
		InputListener input;
		OutputImage image(width, height, "frame");
		image.m_outputDir = outDir;
		World world;
		SimpleCamera camera(&world, &input, image.m_aspectRatio);
		world.ConstructWorld((WorldID)worldId, &camera);
		HomemadeRayTracer tracer(&input, &image, &world);
		tracer.OnInit();                                        // 1-SPP preview mode is the tracer's default (HomemadeRayTracer.h:28)
		using clock = std::chrono::steady_clock;
		for (UINT32 f = 0; f < frames; ++f) {
			const auto &keys = script[f % script.size()];
			for (UINT8 k : keys) { input.NotifyKeyDown(k); input.NotifyKeyDown(k); }     // press + auto-repeat = hold
			camera.OnUpdate(dt);
			for (UINT8 k : keys) input.NotifyKeyUp(k);
			input.Clear();
			const auto t0 = clock::now();
			tracer.TraceRay(&camera, &image);
			const double ms = std::chrono::duration<double, std::milli>(clock::now() - t0).count();
			char name[64]; snprintf(name, sizeof name, "preview_%04u", f);
			image.m_name = name; image.OutputBinary();
			std::cout << "[flythrough] frame " << f << ": 1-SPP preview in " << ms << " ms" << std::endl;
		}
		for (UINT32 done = 0; done < spp;) {
			const UINT32 n = std::min(pass, spp - done);
			const auto t0 = clock::now();
			tracer.TraceRayProgressive(&camera, &image, done, n);
			const double ms = std::chrono::duration<double, std::milli>(clock::now() - t0).count();
			done += n;
			char name[64]; snprintf(name, sizeof name, "refine_%05u", done);
			image.m_name = name; image.OutputBinary();
			std::cout << "[flythrough] " << done << " spp after " << ms << " ms (" << tracer.LastStats().rays << " rays in this pass)" << std::endl;
		}
		tracer.OnDestroy();
		world.DeconstructWorld();
	} catch (const std::exception &e) { std::cerr << "flythrough failed: " << e.what() << std::endl; return 1; }
	return 0;
}
