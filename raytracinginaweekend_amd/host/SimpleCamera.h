// SimpleCamera.h -- look-from/look-at thin-lens camera (reference SimpleCamera.h:8-64).  GetRay runs
// on the device (rt_device.h camera_ray); Export() hands its private frame to the C ABI.
#pragma once
#include "Ray.h"
#include "../../include/mirt.h"

class World;
class InputListener;

class SimpleCamera
{
public:
	SimpleCamera(const World *world, InputListener *inputListener, float aspectRatio);
	void Initialize(const Vec3 &lookFrom, const Vec3 &lookAt, float fov, float minZ, float maxZ, float aperture,
	                float unitsPerSecond, float radiansPerSecond);
	void OnUpdate(float elapsedSeconds);
	void HelpInfo();
	void Export(mirt_camera &out) const;
	BOOL AutoFocus(const Vec3 &lookDir);             // SimpleCamera.cpp:202-214: focus on what the view axis hits

private:
	void Reset();
	void InternalUpdate();

	Vec3  m_origin, m_initialOrigin, m_focus, m_initialFocus, m_vup;
	float m_fov{ 0 }, m_aspectRatio{ 1 }, m_nearPlane{ 0 }, m_farPlane{ 0 }, m_lensRadius{ 0 };
	float m_moveSpeed{ 0 }, m_turnSpeed{ 0 };
	Vec3  m_viewTopLeftCorner, m_viewHorizontal, m_viewVertical, m_u, m_v, m_w;
	const World   *m_world{ nullptr };
	InputListener *m_inputListener{ nullptr };
	float m_yaw{ 0 }, m_pitch{ 0 };
};
