// FlatScene.h -- what the host hands to the C ABI: the pointer graph of IHitable / IMaterial /
// SimpleObjectBVHNode objects flattened into the SoA arrays of mirt_scene (include/mirt.h).
#pragma once
#include "stdafx.h"
#include "../../include/mirt.h"

class IMaterial;
class ITexture2D;

// The instance chain above a leaf hitable, outermost first.  The device record holds exactly
// Translate o RotZ o RotY o RotX o leaf (each optional), the shape SimpleObjectRect/Cube/Sphere
// build (reference SimpleObject.cpp:136,173-181,268-276); other nestings are rejected.
struct InstanceChain
{
	bool  translated = false;
	float off[3] = { 0.0f, 0.0f, 0.0f };
	bool  has_rot = false;
	int   innermost_axis = 3;                 // next RotatedInstance must rotate about a lower axis
	float rot[6] = { 0.0f, 1.0f, 0.0f, 1.0f, 0.0f, 1.0f };   // sinx,cosx,siny,cosy,sinz,cosz
};

class FlatScene
{
public:
	std::vector<uint32_t> prim_kind, prim_axes, prim_mat, obj_first, obj_count, mat_kind;
	std::vector<float>    prim_off, prim_geom, prim_rot, mat_albedo, mat_fuzz, mat_ior, mat_emit, node_min, node_max;
	std::vector<int32_t>  node_left, node_right, mat_texture;
	std::vector<uint32_t> tex_width, tex_height, tex_offset;       // image textures (SimpleTexture2D_TGAImage)
	std::vector<uint8_t>  tex_rgba8;
	int32_t root = MIRT_CHILD_NONE;
	float ambient[3] = { 0.0f, 0.0f, 0.0f };

	std::map<const class Object *, int32_t> objIndex;              // objects already flattened -> their child code

	uint32_t MaterialIndex(const IMaterial *m);                    // interns the material, returns its row
	void AddSphere(const InstanceChain &c, float radius, const IMaterial *m);
	void AddRect(const InstanceChain &c, uint32_t aAxis, uint32_t bAxis, uint32_t cAxis, float a0, float a1, float b0, float b1,
	             float cc, bool reverse, const IMaterial *m);
	void BeginObject() { obj_first.push_back((uint32_t)prim_kind.size()); obj_count.push_back(0); }
	void EndObject();
	mirt_scene View() const;                                       // borrowed pointers into this object

private:
	std::map<const IMaterial *, uint32_t> m_matIndex;
	std::map<const ITexture2D *, int32_t> m_texIndex;
	int32_t TextureIndex(const ITexture2D *t);                     // interns an image texture, returns its row
	void PushCommon(const InstanceChain &c, uint32_t kind, uint32_t axes, const IMaterial *m);
};
