// HomemadeRayTracer.h -- the seam (reference HomemadeRayTracer.h:11-33).  TraceRay flattens the world,
// uploads it through the C ABI of include/mirt.h and renders on the MI355X; there is no CPU path.
#pragma once
#include <vector>
#include "Vec3.h"
#include "../../include/mirt.h"

class OutputImage;
class InputListener;
class SimpleCamera;
class World;

class HomemadeRayTracer
{
public:
	HomemadeRayTracer(InputListener *inputListener, OutputImage *image, const World *world);
	~HomemadeRayTracer();

	void OnInit();
	void OnUpdate(const SimpleCamera *camera, OutputImage *image);
	void OnDestroy();
	void HelpInfo();
	void TraceRay(const SimpleCamera *camera, OutputImage *image);
	// Progressive form for the interactive caller: adds samples [firstSample, firstSample + sampleCount) to the image
	// kept on the device (firstSample == 0 restarts it) and renders the mean of all samples so far into `image`.
	// After passes [0,a) and [a,b) the image equals that of one TraceRay with b samples.
	void TraceRayProgressive(const SimpleCamera *camera, OutputImage *image, UINT32 firstSample, UINT32 sampleCount);

	// SAMPLE_PER_PIXEL / MAX_SAMPLE_DEPTH (reference HomemadeRayTracer.cpp:21-22) at run time, plus what
	// the reference cannot express: a seed, the GPU to use, and the toggles without key events
	void SetSamplesPerPixel(UINT32 spp) { m_spp = spp; }
	void SetMaxDepth(UINT32 depth) { m_maxDepth = depth; }
	void SetSeed(UINT64 seed) { m_seed = seed; }
	void SetDevice(int device) { m_device = device; }
	// Several GPUs of the node for TraceRay: the samples of every pixel are split over them (mirt_multi, include/mirt.h),
	// device ids[0] gathers and resolves.  SetDeviceCount(n) = devices 0..n-1.  Takes effect at the next TraceRay;
	// TraceRayProgressive and the 1-SPP preview stay on one device.
	void SetDevices(const std::vector<int> &ids);
	void SetDeviceCount(int n);
	const mirt_multi_info &LastMultiInfo() const { return m_multiInfo; }
	void Enable1SPP(BOOL on) { m_enable1SPP = on; }
	void EnableNormalDisplay(BOOL on) { m_enableNormalDisplay = on; }
	const mirt_stats &LastStats() const { return m_stats; }

private:
	InputListener *m_inputListener{ nullptr };
	BOOL           m_enableNormalDisplay{ FALSE };
	BOOL           m_enable1SPP{ TRUE };
	const World   *m_world{ nullptr };
	UINT32         m_spp{ 100 }, m_maxDepth{ 50 };
	UINT64         m_seed{ 0x5EED };
	int            m_device{ 0 };
	mirt_ctx      *m_ctx{ nullptr };
	const World   *m_uploadedWorld{ nullptr };
	UINT64         m_uploadedGeneration{ 0 };
	mirt_stats     m_stats{};
	std::vector<int> m_devices;                      // empty: one device (m_device) through m_ctx
	mirt_multi    *m_multi{ nullptr };
	const World   *m_multiWorld{ nullptr };           // what m_multi's devices hold (tracked apart from m_ctx's copy)
	UINT64         m_multiGeneration{ 0 };
	mirt_multi_info m_multiInfo{};
	void ReleaseMulti();
	void Prepare(const SimpleCamera *camera);        // context, scene upload (once per world), camera
};
