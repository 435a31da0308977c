// cornell_example.cpp -- scene code as it would be written against the reference, compiled against the
// drop-in headers: build the Cornell box, press 'M' to leave 1-SPP mode, SPACE to render, 'O' to save.
//   cornell_example [width height spp [outdir]]
#include "stdafx.h"
#include "HomemadeRayTracer.h"
#include "InputListener.h"
#include "OutputImage.h"
#include "SimpleCamera.h"
#include "World.h"

int main(int argc, char **argv)
{
	UINT32 width = argc > 2 ? atoi(argv[1]) : 512, height = argc > 2 ? atoi(argv[2]) : 512, spp = argc > 3 ? atoi(argv[3]) : 100;
	InputListener input;
	OutputImage image(width, height, "OutputImage");
	World world;
	SimpleCamera camera(&world, &input, image.m_aspectRatio);
	world.ConstructWorld(WORLD_ID_CORNELL_BOX, &camera);
	HomemadeRayTracer tracer(&input, &image, &world);
	tracer.OnInit();
	tracer.SetSamplesPerPixel(spp);
	try {
		input.NotifyKeyDown('M'); input.NotifyKeyUp('M'); tracer.OnUpdate(&camera, &image); input.Clear();          // 1-SPP off
		input.NotifyKeyDown(VK_SPACE); input.NotifyKeyUp(VK_SPACE); tracer.OnUpdate(&camera, &image); input.Clear(); // render
	} catch (const std::exception &e) { std::cerr << "render failed: " << e.what() << std::endl; return 1; }
	if (argc > 4) image.m_outputDir = argv[4];
	image.Output();
	tracer.OnDestroy();
	world.DeconstructWorld();
	return 0;
}
