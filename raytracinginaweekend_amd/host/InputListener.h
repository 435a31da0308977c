// InputListener.h -- key state map of the reference (InputListener.h:3-28).  The renderer is headless;
// the class stays because HomemadeRayTracer and SimpleCamera take one and its hot keys toggle modes.
#pragma once
#include "stdafx.h"

class InputListener
{
public:
	void RegisterKey(UINT8 keyCode);
	BOOL WhenPressKey(UINT8 keyCode);
	BOOL WhenReleaseKey(UINT8 keyCode);
	BOOL WhenHoldKey(UINT8 keyCode);
	void NotifyKeyDown(UINT8 keyCode);
	void NotifyKeyUp(UINT8 keyCode);
	void Clear();

private:
	struct KeyState { BOOL down{ FALSE }, press{ FALSE }, release{ FALSE }, hold{ FALSE }; };
	std::map<UINT8, KeyState> m_keyStatus;
};
