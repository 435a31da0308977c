// FlatScene.h -- what the host hands to the C ABI: the pointer graph of IHitable / IMaterial /
// SimpleObjectBVHNode objects flattened into the SoA arrays of mirt_scene (include/mirt.h).
#pragma once
#include "stdafx.h"
#include "../../include/mirt.h"

class IMaterial;
class ITexture2D;

// The instance chain above a leaf hitable, outermost first: any nesting of TranslatedInstance / RotatedInstance, as the
// reference allows (Hitables.h:73-106).  A leaf whose chain is one of the factories' shapes -- Translate(Sphere at 0),
// [Translate(][RotZ(RotY(RotX(]Rect[)))][)] (SimpleObject.cpp:136,173-181,268-276) -- goes to the device in the factory form,
// whose tight loops the list walk keeps; every other leaf in chain form (include/mirt.h "general instance chains").
struct InstanceOp { uint32_t kind; float a, b, c; };      // MIRT_OP_TRANSLATE: offset; MIRT_OP_ROTATE_X/Y/Z: sin, cos, 0
struct InstanceChain
{
	std::vector<InstanceOp> ops;
	bool force_chain = false;          // below a HitableCombo that is not a cube's (six rects inside ONE chain): its leaves take the chain form
};

class FlatScene
{
public:
	std::vector<uint32_t> prim_kind, prim_axes, prim_mat, obj_first, obj_count, mat_kind;
	std::vector<float>    prim_off, prim_geom, prim_rot, mat_albedo, mat_fuzz, mat_ior, mat_emit, node_min, node_max;
	std::vector<int32_t>  node_left, node_right, mat_texture;
	std::vector<uint32_t> tex_width, tex_height, tex_offset;       // image textures (SimpleTexture2D_TGAImage)
	std::vector<uint8_t>  tex_rgba8;
	std::vector<uint32_t> prim_chain, prim_nops, op_kind;          // chain-form primitives (all of prim_chain 0: none)
	std::vector<float>    prim_center, op_arg;
	int32_t root = MIRT_CHILD_NONE;
	float ambient[3] = { 0.0f, 0.0f, 0.0f };

	std::map<const class Object *, int32_t> objIndex;              // objects already flattened -> their child code

	uint32_t MaterialIndex(const IMaterial *m);                    // interns the material, returns its row
	void AddSphere(const InstanceChain &c, const float center[3], float radius, const IMaterial *m);
	void AddRect(const InstanceChain &c, uint32_t aAxis, uint32_t bAxis, uint32_t cAxis, float a0, float a1, float b0, float b1,
	             float cc, bool reverse, const IMaterial *m);
	void BeginObject() { obj_first.push_back((uint32_t)prim_kind.size()); obj_count.push_back(0); }
	void EndObject();
	mirt_scene View() const;                                       // borrowed pointers into this object

private:
	std::map<const IMaterial *, uint32_t> m_matIndex;
	std::map<const ITexture2D *, int32_t> m_texIndex;
	int32_t TextureIndex(const ITexture2D *t);                     // interns an image texture, returns its row
	void PushCommon(uint32_t kind, uint32_t axes, const IMaterial *m, const float off[3], const float rot[6]);
	void PushChain(const InstanceChain &c, const float center[3]);
};
