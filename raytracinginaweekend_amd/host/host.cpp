// host.cpp -- implementation of the drop-in C++ scene API (headers in this directory carry the
// reference's class and method names) on top of the C ABI of include/mirt.h.
//
// What runs here: scene construction (World::ConstructWorld, the object factories, the material
// table), bounding boxes and the SimpleObjectBVHNode build, camera set-up, flattening to mirt_scene,
// and the image sink.  What does not: Hit / Scatter / GetRay / Sample -- those are the HIP kernels.
// Citations are relative to /root/reference/RayTracer/RayTracer/.
#include "stdafx.h"

#include <algorithm>
#include <fstream>
#include <random>
#include <stdexcept>
#include <sys/stat.h>

#include "Hitables.h"
#include "HomemadeRayTracer.h"
#include "InputListener.h"
#include "LightSources.h"
#include "Materials.h"
#include "OutputImage.h"
#include "PPMImageMaker.h"
#include "Randomizer.h"
#include "Resouces.h"
#include "SimpleCamera.h"
#include "SimpleObject.h"
#include "SimpleTexture2D.h"
#include "World.h"

using namespace std;

// =========================================================================================== Randomizer
namespace {
struct HostStream {
	uint32_t key[2]; uint32_t block = 0; uint32_t buf[4]; int left = 0; bool seeded = false;
	void seed(uint64_t s) { key[0] = (uint32_t)s; key[1] = (uint32_t)(s >> 32); block = 0; left = 0; seeded = true; }
	uint32_t next()
	{
		if (!seeded) { std::random_device rd; seed(((uint64_t)rd() << 32) | rd()); }
		if (left == 0) {
			uint32_t c0 = block++, c1 = 0, c2 = 0, c3 = 0x5CE9E, k0 = key[0], k1 = key[1];
			for (int r = 0; r < 10; ++r) {                     // Philox4x32-10
				uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
				uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
				c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
				k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
			}
			buf[0] = c0; buf[1] = c1; buf[2] = c2; buf[3] = c3; left = 4;
		}
		return buf[4 - left--];
	}
} g_stream;
}

void Randomizer::Seed(UINT64 seed) { g_stream.seed(seed); }
UINT32 Randomizer::Raw() { return g_stream.next(); }
float Randomizer::RandomMinMax(float mn, float mx)                       // Randomizer.h:9-17
{
	double r = (double)Raw() / 4294967296.0;
	r = r * (mx - mn) + mn;
	return (float)r;
}
float Randomizer::RandomMinMax2(float mn, float mx)                      // Randomizer.h:20-26
{
	double r = (double)Raw() / 4294967295.0;
	r = r * (mx - mn) + mn;
	return (float)r;
}
Vec3 Randomizer::RomdomInUnitSphere()                                    // Randomizer.h:40-48
{
	Vec3 p;
	do { float x = RandomMinMax2(-1.0f, 1.0f), y = RandomMinMax2(-1.0f, 1.0f), z = RandomMinMax2(-1.0f, 1.0f); p = Vec3(x, y, z); } while (p.squared_length() >= 1.0f);
	return p;
}
Vec3 Randomizer::RandomInUnitDisk()                                      // Randomizer.h:50-58
{
	Vec3 p;
	do { float x = RandomMinMax2(-1.0f, 1.0f), y = RandomMinMax2(-1.0f, 1.0f); p = Vec3(x, y, 0.0f); } while (p.squared_length() >= 1.0f);
	return p;
}

// =========================================================================================== FlatScene
uint32_t FlatScene::MaterialIndex(const IMaterial *m)
{
	if (!m) throw std::runtime_error("FlatScene: hitable without a material");
	auto it = m_matIndex.find(m);
	if (it != m_matIndex.end()) return it->second;
	FlatMaterial f = m->Flatten();
	uint32_t row = (uint32_t)mat_kind.size();
	mat_kind.push_back(f.kind); mat_fuzz.push_back(f.fuzz); mat_ior.push_back(f.ior);
	mat_texture.push_back(f.image ? TextureIndex(f.image) : -1);
	for (int i = 0; i < 3; ++i) { mat_albedo.push_back(f.albedo[i]); mat_emit.push_back(f.emit[i]); }
	m_matIndex[m] = row;
	return row;
}
int32_t FlatScene::TextureIndex(const ITexture2D *t)
{
	auto it = m_texIndex.find(t);
	if (it != m_texIndex.end()) return it->second;
	const auto *img = dynamic_cast<const SimpleTexture2D_TGAImage *>(t);
	if (!img) throw std::runtime_error("FlatScene: only single-colour and TGA image textures are representable");
	if ((uint64_t)tex_rgba8.size() / 4 + (uint64_t)img->m_width * img->m_height > 0xFFFFFFFFull) throw std::runtime_error("FlatScene: textures larger than 2^32 texels");
	int32_t row = (int32_t)tex_width.size();
	tex_width.push_back(img->m_width); tex_height.push_back(img->m_height); tex_offset.push_back((uint32_t)(tex_rgba8.size() / 4));
	tex_rgba8.insert(tex_rgba8.end(), img->m_pixelData.begin(), img->m_pixelData.end());
	m_texIndex[t] = row;
	return row;
}
void FlatScene::PushCommon(uint32_t kind, uint32_t axes, const IMaterial *m, const float off[3], const float rot[6])
{
	if (obj_first.empty()) throw std::runtime_error("FlatScene: primitive outside an object");
	prim_kind.push_back(kind); prim_axes.push_back(axes); prim_mat.push_back(MaterialIndex(m));
	for (int i = 0; i < 3; ++i) prim_off.push_back(off[i]);
	for (int i = 0; i < 6; ++i) prim_rot.push_back(rot[i]);
	prim_chain.push_back(0); prim_nops.push_back(0);
	for (int i = 0; i < 3; ++i) prim_center.push_back(0.0f);
	obj_count.back()++;
}
// turns the primitive pushed last into chain form: its instances, outermost first, and the sphere's own centre
void FlatScene::PushChain(const InstanceChain &c, const float center[3])
{
	if (c.ops.size() > MIRT_MAX_CHAIN_OPS)
		throw std::runtime_error("FlatScene: a leaf hitable inside " + std::to_string(c.ops.size()) + " nested instances; the device takes " + std::to_string(MIRT_MAX_CHAIN_OPS));
	prim_chain.back() = (uint32_t)op_kind.size() + 1u; prim_nops.back() = (uint32_t)c.ops.size();
	for (int i = 0; i < 3; ++i) prim_center[prim_center.size() - 3 + i] = center[i];
	for (const InstanceOp &o : c.ops) { op_kind.push_back(o.kind); op_arg.push_back(o.a); op_arg.push_back(o.b); op_arg.push_back(o.c); }
}
static const float kNoOffset[3] = { 0.0f, 0.0f, 0.0f }, kNoRotation[6] = { 0.0f, 1.0f, 0.0f, 1.0f, 0.0f, 1.0f };
void FlatScene::AddSphere(const InstanceChain &c, const float center[3], float radius, const IMaterial *m)
{
	// SimpleObjectSphere's shape, Translate(Sphere at 0) (SimpleObject.cpp:136): the factory form
	const bool factory = !c.force_chain && c.ops.size() == 1 && c.ops[0].kind == MIRT_OP_TRANSLATE && center[0] == 0.0f && center[1] == 0.0f && center[2] == 0.0f;
	const float off[3] = { factory ? c.ops[0].a : 0.0f, factory ? c.ops[0].b : 0.0f, factory ? c.ops[0].c : 0.0f };
	PushCommon(MIRT_PRIM_SPHERE, 0, m, off, kNoRotation);
	prim_geom.push_back(radius); for (int i = 0; i < 4; ++i) prim_geom.push_back(0.0f);
	if (!factory) PushChain(c, center);
}
void FlatScene::AddRect(const InstanceChain &c, uint32_t aA, uint32_t bA, uint32_t cA, float a0, float a1, float b0, float b1, float cc, bool reverse, const IMaterial *m)
{
	// SimpleObjectRect / SimpleObjectCube's shape, Translate(RotZ(RotY(RotX(Rect)))) (SimpleObject.cpp:173-181), a bare rect, or a
	// rect that is only translated: the factory form
	const size_t n = c.ops.size();
	bool full = n == 4 && c.ops[0].kind == MIRT_OP_TRANSLATE && c.ops[1].kind == MIRT_OP_ROTATE_Z && c.ops[2].kind == MIRT_OP_ROTATE_Y && c.ops[3].kind == MIRT_OP_ROTATE_X;
	const bool factory = !c.force_chain && (full || n == 0 || (n == 1 && c.ops[0].kind == MIRT_OP_TRANSLATE));
	if (!factory) full = false;
	float off[3] = { 0.0f, 0.0f, 0.0f }, rot[6] = { 0.0f, 1.0f, 0.0f, 1.0f, 0.0f, 1.0f };
	if (factory && n) { off[0] = c.ops[0].a; off[1] = c.ops[0].b; off[2] = c.ops[0].c; }
	if (full) { rot[0] = c.ops[3].a; rot[1] = c.ops[3].b; rot[2] = c.ops[2].a; rot[3] = c.ops[2].b; rot[4] = c.ops[1].a; rot[5] = c.ops[1].b; }
	PushCommon(MIRT_PRIM_RECT, MIRT_AXES_PACK(aA, bA, cA, reverse ? 1u : 0u, full ? 1u : 0u), m, off, rot);
	prim_geom.push_back(a0); prim_geom.push_back(a1); prim_geom.push_back(b0); prim_geom.push_back(b1); prim_geom.push_back(cc);
	if (!factory) PushChain(c, kNoOffset);
}
void FlatScene::EndObject()
{
	if (obj_count.back() == 0) throw std::runtime_error("FlatScene: object without primitives");
}
mirt_scene FlatScene::View() const
{
	mirt_scene s{};
	s.struct_size = (uint32_t)sizeof s;
	s.n_prims = (uint32_t)prim_kind.size(); s.prim_kind = prim_kind.data(); s.prim_off = prim_off.data(); s.prim_geom = prim_geom.data();
	s.prim_axes = prim_axes.data(); s.prim_rot = prim_rot.data(); s.prim_mat = prim_mat.data();
	s.n_objects = (uint32_t)obj_first.size(); s.obj_first = obj_first.data(); s.obj_count = obj_count.data();
	s.n_mats = (uint32_t)mat_kind.size(); s.mat_kind = mat_kind.data(); s.mat_albedo = mat_albedo.data(); s.mat_fuzz = mat_fuzz.data();
	s.mat_ior = mat_ior.data(); s.mat_emit = mat_emit.data();
	s.n_nodes = (uint32_t)node_left.size(); s.node_min = node_min.data(); s.node_max = node_max.data();
	s.node_left = node_left.data(); s.node_right = node_right.data(); s.root = root;
	for (int i = 0; i < 3; ++i) s.ambient[i] = ambient[i];
	s.n_textures = (uint32_t)tex_width.size(); s.tex_width = tex_width.data(); s.tex_height = tex_height.data(); s.tex_offset = tex_offset.data();
	s.tex_rgba8 = tex_rgba8.data(); s.mat_texture = mat_texture.data();
	bool chains = false; for (uint32_t k : prim_chain) chains |= k != 0;
	if (chains) { s.prim_chain = prim_chain.data(); s.prim_nops = prim_nops.data(); s.prim_center = prim_center.data(); s.n_ops = (uint32_t)op_kind.size(); s.op_kind = op_kind.data(); s.op_arg = op_arg.data(); }
	return s;
}

// =========================================================================================== Hitables
AABB SphereHitable::BoundingBox() const                                  // Hitables.cpp:56-60
{
	Vec3 r(m_radius, m_radius, m_radius);
	return AABB(-r, r);
}
void SphereHitable::Flatten(FlatScene &out, const InstanceChain &chain) const
{
	const float c[3] = { m_center.x(), m_center.y(), m_center.z() };   // its own centre stays its own (Hitables.cpp:20,37,65): not a translation
	out.AddSphere(chain, c, m_radius, m_material);
}

AxisAlignedRectHitable::AxisAlignedRectHitable(UINT32 aAxisIndex, UINT32 bAxisIndex, float a0, float a1, float b0, float b1, float c, BOOL reverseFace)
	: m_a0(a0), m_a1(a1), m_b0(b0), m_b1(b1), m_c(c), m_aAxisIndex(aAxisIndex), m_bAxisIndex(bAxisIndex), m_reverseFace(reverseFace)
{
	if (!(aAxisIndex < 3 && bAxisIndex < 3 && aAxisIndex != bAxisIndex)) throw std::invalid_argument("AxisAlignedRectHitable: invalid axes");
	m_cAxisIndex = 3 - aAxisIndex - bAxisIndex;                          // the remaining axis (Hitables.cpp:84)
	if (m_a0 > m_a1) std::swap(m_a0, m_a1);
	if (m_b0 > m_b1) std::swap(m_b0, m_b1);
}
AABB AxisAlignedRectHitable::BoundingBox() const                         // Hitables.cpp:117-128
{
	Vec3 mn(0, 0, 0), mx(0, 0, 0);
	mn.set(m_aAxisIndex, m_a0); mx.set(m_aAxisIndex, m_a1);
	mn.set(m_bAxisIndex, m_b0); mx.set(m_bAxisIndex, m_b1);
	mn.set(m_cAxisIndex, m_c - 0.0001f); mx.set(m_cAxisIndex, m_c + 0.0001f);
	return AABB(mn, mx);
}
void AxisAlignedRectHitable::Flatten(FlatScene &out, const InstanceChain &chain) const
{
	out.AddRect(chain, m_aAxisIndex, m_bAxisIndex, m_cAxisIndex, m_a0, m_a1, m_b0, m_b1, m_c, m_reverseFace != FALSE, m_material);
}

void HitableCombo::BindMaterial(IMaterial *m) { for (UINT32 i = 0; i < m_hitableCount; ++i) m_hitableList[i]->BindMaterial(m); }
AABB HitableCombo::BoundingBox() const                                   // Hitables.cpp:152-161
{
	if (!m_hitableCount) throw std::runtime_error("HitableCombo: empty");
	AABB box = m_hitableList[0]->BoundingBox();
	for (UINT32 i = 1; i < m_hitableCount; ++i) box = CombineAABB(m_hitableList[i]->BoundingBox(), box);
	return box;
}
void HitableCombo::Flatten(FlatScene &out, const InstanceChain &chain) const
{
	// HitableCombo::Hit (Hitables.cpp:130-142) tries its children in order with t_max shrinking, which is what an object's run of
	// primitives does on the device; the instances above the combo apply to every child, nested combos unfold in place
	// A combo of rects alone is a cube's (SimpleObject.cpp:261-276): its faces share the chain and may keep the factory form, whose
	// object box is built combo first, chain second, like the reference's.  Any other combo hands its leaves over in chain form.
	InstanceChain c = chain;
	for (UINT32 i = 0; i < m_hitableCount; ++i) if (!dynamic_cast<AxisAlignedRectHitable *>(m_hitableList[i])) c.force_chain = true;
	for (UINT32 i = 0; i < m_hitableCount; ++i) m_hitableList[i]->Flatten(out, c);
}

AABB TranslatedInstance::BoundingBox() const                             // Hitables.cpp:175-180
{
	AABB b = m_hitable->BoundingBox();
	return AABB(b.m_min + m_offset, b.m_max + m_offset);
}
void TranslatedInstance::Flatten(FlatScene &out, const InstanceChain &chain) const
{
	InstanceChain c = chain;
	c.ops.push_back(InstanceOp{ MIRT_OP_TRANSLATE, m_offset.x(), m_offset.y(), m_offset.z() });
	m_hitable->Flatten(out, c);
}

RotatedInstance::RotatedInstance(IHitable *hitable, float angle, UINT32 rotateAxis)    // Hitables.cpp:182-239
	: TransformedInstance(hitable), m_sinTheta(sinf(angle)), m_cosTheta(cosf(angle)), m_cAxisIndex(rotateAxis)
{
	if (rotateAxis == 0) { m_aAxisIndex = 1; m_bAxisIndex = 2; m_OP = -1.0f; }
	else if (rotateAxis == 1) { m_aAxisIndex = 0; m_bAxisIndex = 2; m_OP = 1.0f; }
	else { m_cAxisIndex = 2; m_aAxisIndex = 0; m_bAxisIndex = 1; m_OP = -1.0f; }
	m_oppositeOP = -m_OP;
	AABB in = m_hitable->BoundingBox();
	Vec3 mn(FLT_MAX, FLT_MAX, FLT_MAX), mx(-FLT_MAX, -FLT_MAX, -FLT_MAX);
	for (UINT32 corner = 0; corner < 8; ++corner) {                      // i,j,k of the reference = bits 2,1,0
		UINT32 i = (corner >> 2) & 1, j = (corner >> 1) & 1, k = corner & 1;
		float a = i * in.m_max[m_aAxisIndex] + (1 - i) * in.m_min[m_aAxisIndex];
		float b = j * in.m_max[m_bAxisIndex] + (1 - j) * in.m_min[m_bAxisIndex];
		float c = k * in.m_max[m_cAxisIndex] + (1 - k) * in.m_min[m_cAxisIndex];
		Vec3 t(0, 0, 0);
		t.set(m_aAxisIndex, m_cosTheta * a + m_OP * m_sinTheta * b);
		t.set(m_bAxisIndex, -m_OP * m_sinTheta * a + m_cosTheta * b);
		t.set(m_cAxisIndex, c);
		for (UINT32 ax = 0; ax < 3; ++ax) { if (t[ax] > mx[ax]) mx.set(ax, t[ax]); if (t[ax] < mn[ax]) mn.set(ax, t[ax]); }
	}
	m_boundingBox = AABB(mn, mx);
}
void RotatedInstance::Flatten(FlatScene &out, const InstanceChain &chain) const
{
	InstanceChain c = chain;
	c.ops.push_back(InstanceOp{ MIRT_OP_ROTATE_X + m_cAxisIndex, m_sinTheta, m_cosTheta, 0.0f });
	m_hitable->Flatten(out, c);
}

// =========================================================================================== Materials
// the albedo texture of a material: a single colour goes into the material row, an image is referenced
static void albedo_texture(const ITexture2D *t, FlatMaterial &f)
{
	if (!t) throw std::runtime_error("material without a texture");
	if (t->IsSingleColor()) { Vec3 c = t->Sample(0.0f, 0.0f); f.albedo[0] = c.r(); f.albedo[1] = c.g(); f.albedo[2] = c.b(); }
	else f.image = t;
}
FlatMaterial Lambertian::Flatten() const { FlatMaterial f{}; f.kind = MIRT_MAT_LAMBERTIAN; albedo_texture(m_albedo, f); return f; }
Metal::Metal(ITexture2D *albedo, float fuzziness) : m_albedo(albedo)     // Materials.cpp:93-97
{
	m_fuzziness = (fuzziness < 1.0f) ? ((fuzziness >= 0.0f) ? fuzziness : 0.0f) : 1.0f;
}
FlatMaterial Metal::Flatten() const { FlatMaterial f{}; f.kind = MIRT_MAT_METAL; albedo_texture(m_albedo, f); f.fuzz = m_fuzziness; return f; }

// =========================================================================================== image textures
void DecodeTGA(const std::vector<UINT8> &file, UINT32 &width, UINT32 &height, std::vector<UINT8> &rgba)
{
	auto bad = [](const char *why) { throw std::runtime_error(std::string("DecodeTGA: ") + why); };
	if (file.size() < 18) bad("file shorter than the header");
	const UINT32 idLength = file[0], type = file[2];
	const UINT32 cmapOrigin = file[3] | (file[4] << 8), cmapLength = file[5] | (file[6] << 8), cmapDepth = file[7];
	width = file[12] | (file[13] << 8); height = file[14] | (file[15] << 8);
	const UINT32 depth = file[16], descriptor = file[17];
	if (!width || !height) bad("empty image");
	const bool rle = type == 9 || type == 10 || type == 11;
	const UINT32 base = rle ? type - 8 : type;
	if (base < 1 || base > 3) bad("unsupported image type");
	if (base == 1 && (depth != 8 || (cmapDepth != 24 && cmapDepth != 32))) bad("colour-mapped images need 8-bit indices and a 24/32-bit palette");
	if (base == 2 && depth != 24 && depth != 32) bad("true-colour images must be 24 or 32 bits");
	if (base == 3 && depth != 8) bad("grey images must be 8 bits");
	const size_t palette = 18 + (size_t)idLength;
	size_t at = palette + (size_t)(cmapDepth / 8) * cmapLength;
	const UINT32 bpp = depth / 8;
	const size_t count = (size_t)width * height;
	std::vector<UINT8> raw(count * bpp);
	if (!rle) {
		if (at + raw.size() > file.size()) bad("truncated pixel data");
		memcpy(raw.data(), file.data() + at, raw.size());
	} else {
		size_t done = 0;
		while (done < count) {
			if (at >= file.size()) bad("truncated RLE stream");
			const UINT32 packet = file[at++], n = (packet & 0x7F) + 1;
			if (done + n > count) bad("RLE packet runs past the image");
			if (packet & 0x80) {
				if (at + bpp > file.size()) bad("truncated RLE stream");
				for (UINT32 k = 0; k < n; ++k) memcpy(raw.data() + (done + k) * bpp, file.data() + at, bpp);
				at += bpp;
			} else {
				if (at + (size_t)n * bpp > file.size()) bad("truncated RLE stream");
				memcpy(raw.data() + done * bpp, file.data() + at, (size_t)n * bpp);
				at += (size_t)n * bpp;
			}
			done += n;
		}
	}
	rgba.assign(count * 4, 255);
	const bool right = descriptor & 0x10, upper = descriptor & 0x20;
	for (UINT32 i = 0; i < height; ++i)
		for (UINT32 j = 0; j < width; ++j) {
			const UINT8 *px = raw.data() + ((size_t)i * width + j) * bpp;
			UINT8 r, g, b, a = 255;
			if (base == 2) { b = px[0]; g = px[1]; r = px[2]; if (bpp == 4) a = px[3]; }
			else if (base == 3) { r = g = b = px[0]; }
			else {
				const int idx = (int)px[0] - (int)cmapOrigin;
				if (idx < 0 || (UINT32)idx >= cmapLength) { r = g = b = a = 255; }
				else {
					const size_t e = palette + (size_t)idx * (cmapDepth / 8);
					if (e + cmapDepth / 8 > file.size()) bad("truncated palette");
					b = file[e]; g = file[e + 1]; r = file[e + 2]; if (cmapDepth == 32) a = file[e + 3];
				}
			}
			const UINT32 y = upper ? i : height - 1 - i, x = right ? width - 1 - j : j;
			UINT8 *o = rgba.data() + ((size_t)y * width + x) * 4;
			o[0] = r; o[1] = g; o[2] = b; o[3] = a;
		}
}
SimpleTexture2D_TGAImage::SimpleTexture2D_TGAImage(const char *filePath)
{
	std::ifstream f(filePath, std::ios::binary);
	if (!f) throw std::runtime_error(std::string("SimpleTexture2D_TGAImage: cannot open ") + filePath);
	std::vector<UINT8> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	DecodeTGA(file, m_width, m_height, m_pixelData);
}
SimpleTexture2D_TGAImage::SimpleTexture2D_TGAImage(UINT32 width, UINT32 height, const UINT8 *rgba)
{
	if (!width || !height || !rgba) throw std::invalid_argument("SimpleTexture2D_TGAImage: empty image");
	m_width = width; m_height = height; m_pixelData.assign(rgba, rgba + (size_t)width * height * 4);
}
Vec3 SimpleTexture2D_TGAImage::Sample(float u, float v) const             // SimpleTexture2D.cpp:130-147 (the render path samples on the device)
{
	auto toU32 = [](float f) -> UINT32 { if (!(f > -9223372036854775808.0f && f < 9223372036854775808.0f)) return 0u; return (UINT32)(UINT64)(long long)f; };
	UINT32 i = toU32(u * m_width), j = toU32((1.0f - v) * m_height - 0.001f);
	if (i > m_width - 1) i = m_width - 1;
	if (j > m_height - 1) j = m_height - 1;
	const UINT8 *px = m_pixelData.data() + ((size_t)j * m_width + i) * 4;
	return Vec3(px[0] / 255.0f, px[1] / 255.0f, px[2] / 255.0f);
}
FlatMaterial Dielectric::Flatten() const { FlatMaterial f{}; f.kind = MIRT_MAT_DIELECTRIC; f.ior = m_refractiveIndex; return f; }
FlatMaterial DiffuseLight::Flatten() const
{
	FlatMaterial f{}; f.kind = MIRT_MAT_DIFFUSE_LIGHT;
	f.emit[0] = m_intensity.x(); f.emit[1] = m_intensity.y(); f.emit[2] = m_intensity.z();
	return f;
}

// =========================================================================================== Objects
Object::~Object() { delete m_hitable; m_hitable = nullptr; }
AABB Object::BoundingBox() const { return m_hitable->BoundingBox(); }
int32_t Object::Flatten(FlatScene &out) const
{
	auto seen = out.objIndex.find(this);                 // World::Flatten lists the objects in creation order first
	if (seen != out.objIndex.end()) return seen->second;
	out.objIndex[this] = ~(int32_t)out.obj_first.size();
	out.BeginObject();
	m_hitable->Flatten(out, InstanceChain{});
	out.EndObject();
	return ~(int32_t)(out.obj_first.size() - 1);
}

SimpleObjectSphere::SimpleObjectSphere(const Vec3 &center, float radius, Mesh *mesh, IMaterial *material, World *world)   // SimpleObject.cpp:129-139
{
	m_translation = center; m_scaling = Vec3(radius, radius, radius); m_rotation = Vec3(0.0f, 0.0f, 0.0f);
	m_mesh = mesh; m_material = material; m_world = world;
	m_hitable = new TranslatedInstance(new SphereHitable(Vec3(0.0f, 0.0f, 0.0f), radius), m_translation);
	m_hitable->BindMaterial(material);
}

static IHitable *wrap_chain(IHitable *leaf, const Vec3 &rotation, const Vec3 &translation)   // SimpleObject.cpp:173-181
{
	return new TranslatedInstance(new RotatedInstance(new RotatedInstance(new RotatedInstance(leaf, rotation[0], 0), rotation[1], 1), rotation[2], 2), translation);
}

SimpleObjectRect::SimpleObjectRect(SimpleObjectRectAlignAxes axes, const Vec3 &center, const Vec3 &rotation, float width, float height, BOOL reverseFace,
                                   Mesh *mesh, IMaterial *material, World *world)                 // SimpleObject.cpp:141-184
	: m_alignAxes(axes), m_reverseFace(reverseFace)
{
	m_translation = center; m_scaling = Vec3(width, height, 1.0f); m_rotation = rotation;
	m_mesh = mesh; m_material = material; m_world = world;
	UINT32 aA = (axes == ZY_RECT) ? 2 : 0, bA = (axes == XZ_RECT) ? 2 : 1;
	float ha = width / 2.0f, hb = height / 2.0f;
	m_hitable = wrap_chain(new AxisAlignedRectHitable(aA, bA, -ha, ha, -hb, hb, 0.0f, m_reverseFace), m_rotation, m_translation);
	m_hitable->BindMaterial(material);
}

SimpleObjectCube::SimpleObjectCube(const Vec3 &center, const Vec3 &rotation, const Vec3 &size, Mesh *mesh, IMaterial *material, World *world)   // SimpleObject.cpp:249-279
{
	m_translation = center; m_scaling = size; m_rotation = rotation; m_mesh = mesh; m_material = material; m_world = world;
	Vec3 lo = -size * 0.5f, hi = size * 0.5f;
	struct F { UINT32 a, b; float a0, a1, b0, b1, c; BOOL rev; } faces[6] = {
		{ 0, 2, lo.x(), hi.x(), lo.z(), hi.z(), hi.y(), FALSE }, { 0, 2, lo.x(), hi.x(), lo.z(), hi.z(), lo.y(), TRUE },
		{ 0, 1, lo.x(), hi.x(), lo.y(), hi.y(), hi.z(), FALSE }, { 0, 1, lo.x(), hi.x(), lo.y(), hi.y(), lo.z(), TRUE },
		{ 2, 1, lo.z(), hi.z(), lo.y(), hi.y(), hi.x(), FALSE }, { 2, 1, lo.z(), hi.z(), lo.y(), hi.y(), lo.x(), TRUE } };
	for (int i = 0; i < 6; ++i) m_faceList[i] = new AxisAlignedRectHitable(faces[i].a, faces[i].b, faces[i].a0, faces[i].a1, faces[i].b0, faces[i].b1, faces[i].c, faces[i].rev);
	m_hitable = wrap_chain(new HitableCombo(&m_faceList[0], 6), m_rotation, m_translation);
	m_hitable->BindMaterial(material);
}
SimpleObjectCube::~SimpleObjectCube()
{
	for (auto &f : m_faceList) { delete f; f = nullptr; }
}

// SimpleObject.cpp:293-346: random axis, descending by translation, split at floor(n/2).  stable_sort
// pins the order of ties, which std::sort leaves unspecified.
SimpleObjectBVHNode::SimpleObjectBVHNode(std::vector<Object *> objects)
{
	if (objects.empty()) throw std::invalid_argument("SimpleObjectBVHNode: no objects");
	UINT32 axis = UINT32(3 * Randomizer::RandomUNorm());
	if (axis > 2) axis = 2;
	std::stable_sort(objects.begin(), objects.end(), [axis](Object *a, Object *b) { return b->m_translation[axis] < a->m_translation[axis]; });
	size_t n = objects.size();
	if (n == 1) { leftChild = objects[0]; rightChild = nullptr; m_bindingBox = leftChild->BoundingBox(); return; }
	if (n == 2) { leftChild = objects[0]; rightChild = objects[1]; }
	else {
		size_t half = static_cast<size_t>(n * 0.5f);
		leftChild = new SimpleObjectBVHNode(std::vector<Object *>(objects.begin(), objects.begin() + half));
		rightChild = new SimpleObjectBVHNode(std::vector<Object *>(objects.begin() + half, objects.end()));
	}
	m_bindingBox = CombineAABB(leftChild->BoundingBox(), rightChild->BoundingBox());
}
SimpleObjectBVHNode::~SimpleObjectBVHNode() { delete leftChild; delete rightChild; leftChild = rightChild = nullptr; }
int32_t SimpleObjectBVHNode::Flatten(FlatScene &out) const
{
	int32_t me = (int32_t)out.node_left.size();
	out.node_left.push_back(MIRT_CHILD_NONE); out.node_right.push_back(MIRT_CHILD_NONE);
	for (int i = 0; i < 3; ++i) { out.node_min.push_back(m_bindingBox.m_min[i]); out.node_max.push_back(m_bindingBox.m_max[i]); }
	int32_t l = leftChild ? leftChild->Flatten(out) : MIRT_CHILD_NONE;
	int32_t r = rightChild ? rightChild->Flatten(out) : MIRT_CHILD_NONE;
	out.node_left[me] = l; out.node_right[me] = r;
	return me;
}

// =========================================================================================== Resources
void Resources::Load() { LoadMaterials(); }
void Resources::Unload()
{
	for (auto *m : m_materials) delete m;
	for (auto *t : m_textures) delete t;
	m_materials.clear(); m_textures.clear();
}
IMaterial *Resources::GetTheMaterial(MaterialUniqueID id) const
{
	if ((size_t)id >= m_materials.size()) throw std::out_of_range("Resources::GetTheMaterial");
	return m_materials[id];
}
void Resources::LoadMaterials()                                          // Resouces.cpp:104-209, same order = same ids
{
	auto colour = [&](float r, float g, float b) { ITexture2D *t = new SimpleTexture2D_SingleColor(Vec3(r, g, b)); m_textures.push_back(t); return t; };
	for (int i = 0; i < MATERIAL_ID_RANDOM_LAMBERTIAN_COUNT; ++i) {
		float c[3];
		for (int k = 0; k < 3; ++k) { float u0 = Randomizer::RandomUNorm(); float u1 = Randomizer::RandomUNorm(); c[k] = u0 * u1; }
		m_materials.push_back(new Lambertian(colour(c[0], c[1], c[2])));
	}
	for (int i = 0; i < MATERIAL_ID_RANDOM_METAL_COUNT; ++i) {
		float c[3];
		for (int k = 0; k < 3; ++k) c[k] = 0.5f * (1 + Randomizer::RandomUNorm());
		float fuzz = 0.5f * Randomizer::RandomUNorm();
		m_materials.push_back(new Metal(colour(c[0], c[1], c[2]), fuzz));
	}
	m_materials.push_back(new Lambertian(colour(0.5f, 0.5f, 0.5f)));      // MATERIAL_ID_LAMBERTIAN0
	m_materials.push_back(new Lambertian(colour(0.4f, 0.2f, 0.1f)));      // 1
	m_materials.push_back(new Lambertian(colour(0.65f, 0.05f, 0.05f)));   // 2
	m_materials.push_back(new Lambertian(colour(0.73f, 0.73f, 0.73f)));   // 3
	m_materials.push_back(new Lambertian(colour(0.12f, 0.45f, 0.15f)));   // 4
	m_materials.push_back(new Metal(colour(0.7f, 0.6f, 0.5f), 0.0f));     // MATERIAL_ID_METAL
	m_materials.push_back(new Dielectric(1.5f));                          // MATERIAL_ID_DIELECTRIC
	// the two TGA-textured materials (Resouces.cpp:166-176; unused by the shipped worlds).  The reference reads
	// ..\\Assets\\*.tga relative to its working directory; here the directory comes from $MIRT_ASSETS (default
	// ../Assets), and a file that is not there (one of the two is missing from the reference repository itself)
	// leaves a mid-grey stand-in so the later ids keep their values
	auto image = [&](const char *name) -> ITexture2D * {
		const char *dir = getenv("MIRT_ASSETS");
		std::string path = std::string(dir ? dir : "../Assets") + "/" + name;
		struct stat st;
		if (stat(path.c_str(), &st) != 0) return colour(0.5f, 0.5f, 0.5f);
		ITexture2D *t = new SimpleTexture2D_TGAImage(path.c_str());
		m_textures.push_back(t);
		return t;
	};
	m_materials.push_back(new Lambertian(image("pab_ground_soil_001_c.tga")));          // MATERIAL_ID_IMAGE_BASED_GROUND_SOIL
	m_materials.push_back(new Metal(image("pro_metal_checker_plate_001_c.tga"), 0.7f)); // MATERIAL_ID_IMAGE_BASED_METAL_CHECKER
	m_materials.push_back(new DiffuseLight(Vec3(4.0f, 4.0f, 4.0f)));      // WHITE
	m_materials.push_back(new DiffuseLight(Vec3(15.0f, 15.0f, 15.0f)));   // BRIGHT
	m_materials.push_back(new DiffuseLight(Vec3(4.0f, 0.0f, 0.0f)));      // RED
	m_materials.push_back(new DiffuseLight(Vec3(0.0f, 4.0f, 0.0f)));      // GREEN
	m_materials.push_back(new DiffuseLight(Vec3(0.0f, 0.0f, 4.0f)));      // BLUE
	m_materials.push_back(new Lambertian(colour(0.19f, 0.2f, 0.14f)));    // MATERIAL_ID_CORNELL_BALL1
	m_materials.push_back(new Metal(colour(0.26f, 0.06f, 0.06f), 0.20f)); // MATERIAL_ID_CORNELL_BALL2
}

// =========================================================================================== LightSources
LightSources::LightSources(World *world, std::vector<Object *> &objects, const Vec3 &ambientLight)   // LightSources.cpp:14-30
	: m_world(world), m_ambientLight(ambientLight)
{
	for (Object *o : objects) if (o->m_material && o->m_material->GetID() == MID_DIFFUSE_LIGHT) m_lightSources.push_back(o);
	m_lightSourceCount = (UINT32)m_lightSources.size();
	m_lightSourceCountX = m_lightSourceCount == 0 ? 1 : m_lightSourceCount;
}

// =========================================================================================== World
void World::ConstructWorld(WorldID wid, SimpleCamera *camera)            // World.cpp:15-110
{
	++m_generation;
	m_resources = new Resources();
	m_resources->Load();
	auto mat = [&](MaterialUniqueID id) { return m_resources->GetTheMaterial(id); };
	std::vector<Object *> objects;
	const Vec3 none(0.0f, 0.0f, 0.0f);
	if (wid == WORLD_ID_RANDOM_SPHERES || wid == WORLD_ID_FOUR_SPHERES) {
		objects.push_back(new SimpleObjectSphere(Vec3(0.0f, -1000.0f, -0.0f), 1000.0f, nullptr, mat(MATERIAL_ID_LAMBERTIAN0), this));
		objects.push_back(new SimpleObjectSphere(Vec3(-4.0f, 1.0f, 0.0f), 1.0f, nullptr, mat(MATERIAL_ID_LAMBERTIAN1), this));
		objects.push_back(new SimpleObjectSphere(Vec3(0.0f, 1.0f, 0.0f), 1.0f, nullptr, mat(MATERIAL_ID_DIELECTRIC), this));
		objects.push_back(new SimpleObjectSphere(Vec3(4.0f, 1.0f, 0.0f), 1.0f, nullptr, mat(MATERIAL_ID_METAL), this));
		if (wid == WORLD_ID_RANDOM_SPHERES) {
			for (int a = -11; a < 11; a++) for (int b = -11; b < 11; b++) {
				float chooseMat = Randomizer::RandomUNorm();
				float cx = a + 0.9f * Randomizer::RandomUNorm();
				float cz = b + 0.9f * Randomizer::RandomUNorm();
				Vec3 center(cx, 0.2f, cz);
				if ((center - Vec3(4.0f, 0.2f, 0.0f)).length() > 0.9f) {
					MaterialUniqueID id = MATERIAL_ID_DIELECTRIC;
					if (chooseMat < 0.8f) id = (MaterialUniqueID)(UINT32)(MATERIAL_ID_RANDOM_LAMBERTIAN_START + Randomizer::RandomUNorm() * MATERIAL_ID_RANDOM_LAMBERTIAN_COUNT);
					else if (chooseMat < 0.95f) id = (MaterialUniqueID)(UINT32)(MATERIAL_ID_RANDOM_METAL_START + Randomizer::RandomUNorm() * MATERIAL_ID_RANDOM_METAL_COUNT);
					objects.push_back(new SimpleObjectSphere(center, 0.2f, nullptr, mat(id), this));
				}
			}
		}
		m_lightSources = new LightSources(this, objects, Vec3(0.85f, 0.9f, 1.0f));
		camera->Initialize(Vec3(11.0f, 2.0f, 3.0f), Vec3(0.0f, 0.6f, 0.0f), 20.0f, 1.0f, 10000.0f, 0.1f, 10.0f, 1.0f);
	} else if (wid == WORLD_ID_CORNELL_BOX || wid == WORLD_ID_CORNELL_TWO_BOXES) {
		objects.push_back(new SimpleObjectRect(XZ_RECT, Vec3(0.0f, 0.99f, 0.0f), none, 0.65f, 0.65f, TRUE, nullptr, mat(MATERIAL_ID_LIGHTSOURCE_BRIGHT), this));
		objects.push_back(new SimpleObjectRect(ZY_RECT, Vec3(1.0f, 0.0f, 0.0f), none, 2.0f, 2.0f, TRUE, nullptr, mat(MATERIAL_ID_LAMBERTIAN4), this));
		objects.push_back(new SimpleObjectRect(ZY_RECT, Vec3(-1.0f, 0.0f, 0.0f), none, 2.0f, 2.0f, FALSE, nullptr, mat(MATERIAL_ID_LAMBERTIAN2), this));
		objects.push_back(new SimpleObjectRect(XY_RECT, Vec3(0.0f, 0.0f, -1.0f), none, 2.0f, 2.0f, FALSE, nullptr, mat(MATERIAL_ID_LAMBERTIAN3), this));
		objects.push_back(new SimpleObjectRect(XZ_RECT, Vec3(0.0f, 1.0f, 0.0f), none, 2.0f, 2.0f, TRUE, nullptr, mat(MATERIAL_ID_LAMBERTIAN3), this));
		objects.push_back(new SimpleObjectRect(XZ_RECT, Vec3(0.0f, -1.0f, 0.0f), none, 2.0f, 2.0f, FALSE, nullptr, mat(MATERIAL_ID_LAMBERTIAN3), this));
		if (wid == WORLD_ID_CORNELL_TWO_BOXES) {                             // World.cpp:84-89
			float height = 1.2f;
			objects.push_back(new SimpleObjectCube(Vec3(-0.3f, -1.0f + height * 0.5f, -0.4f), Vec3(0.0f, 0.3f, 0.0f), Vec3(0.6f, height, 0.6f), nullptr, mat(MATERIAL_ID_LAMBERTIAN3), this));
			height = 0.6f;
			objects.push_back(new SimpleObjectCube(Vec3(0.32f, -1.0f + height * 0.5f, 0.4f), Vec3(0.0f, -0.3f, 0.0f), Vec3(height, height, height), nullptr, mat(MATERIAL_ID_LAMBERTIAN3), this));
			camera->Initialize(Vec3(0.0f, 0.0f, 4.0f), Vec3(0.0f, 0.0f, 0.0f), 40.0f, 1.0f, 10000.0f, 0.0f, 10.0f, 1.0f);
		} else {                                                             // World.cpp:91-96
			float r = 0.32f;
			objects.push_back(new SimpleObjectSphere(Vec3(-0.38f, -1.0f + r, -0.35f), r, nullptr, mat(MATERIAL_ID_CORNELL_BALL1), this));
			objects.push_back(new SimpleObjectSphere(Vec3(0.0f, -1.0f + r, 0.50f), r, nullptr, mat(MATERIAL_ID_DIELECTRIC), this));
			objects.push_back(new SimpleObjectSphere(Vec3(0.38f, -1.0f + r, -0.35f), r, nullptr, mat(MATERIAL_ID_CORNELL_BALL2), this));
			camera->Initialize(Vec3(0.0f, 0.64f, 4.63f), Vec3(0.0f, -0.26f, -1.0f), 30.0f, 1.0f, 10000.0f, 0.0f, 10.0f, 1.0f);
		}
		m_lightSources = new LightSources(this, objects, Vec3(0.0f, 0.0f, 0.0f));
	} else if (wid == WORLD_ID_TEXTURED) {
		objects.push_back(new SimpleObjectRect(XZ_RECT, Vec3(0.0f, 0.0f, 0.0f), none, 4.0f, 4.0f, FALSE, nullptr, mat(MATERIAL_ID_IMAGE_BASED_GROUND_SOIL), this));
		objects.push_back(new SimpleObjectRect(XY_RECT, Vec3(0.0f, 1.5f, -2.0f), none, 4.0f, 3.0f, FALSE, nullptr, mat(MATERIAL_ID_IMAGE_BASED_GROUND_SOIL), this));
		objects.push_back(new SimpleObjectSphere(Vec3(0.0f, 0.7f, 0.0f), 0.7f, nullptr, mat(MATERIAL_ID_IMAGE_BASED_METAL_CHECKER), this));
		objects.push_back(new SimpleObjectRect(XZ_RECT, Vec3(0.0f, 3.0f, 0.0f), none, 1.5f, 1.5f, TRUE, nullptr, mat(MATERIAL_ID_LIGHTSOURCE_WHITE), this));
		camera->Initialize(Vec3(0.0f, 1.5f, 5.0f), Vec3(0.0f, 0.7f, 0.0f), 35.0f, 1.0f, 10000.0f, 0.0f, 10.0f, 1.0f);
		m_lightSources = new LightSources(this, objects, Vec3(0.5f, 0.6f, 0.8f));
	} else if (wid == WORLD_ID_CHAINS) {
		// scene code written against the reference's Hitables.h, no factory in sight: plain Objects whose m_hitable is any
		// nesting of instances and combos (the combos' child arrays and children are owned by the object, as SimpleObjectCube
		// owns its faces)
		struct Built : Object {
			std::vector<IHitable *> owned; std::vector<IHitable **> lists;
			~Built() override { for (IHitable *h : owned) delete h; for (IHitable **l : lists) delete[] l; }
			IHitable *combo(std::initializer_list<IHitable *> children) {
				IHitable **l = new IHitable *[children.size()]; size_t n = 0;
				for (IHitable *h : children) { l[n++] = h; owned.push_back(h); }
				lists.push_back(l);
				return new HitableCombo(l, (UINT32)children.size());
			}
		};
		auto add = [&](Built *o, IHitable *h, const Vec3 &where, MaterialUniqueID id) {
			o->m_world = this; o->m_translation = where; o->m_scaling = Vec3(1.0f, 1.0f, 1.0f); o->m_rotation = none;
			o->m_hitable = h; o->m_material = mat(id); h->BindMaterial(o->m_material);
			objects.push_back(o);
		};
		objects.push_back(new SimpleObjectRect(XZ_RECT, Vec3(0.0f, -2.0f, 0.0f), none, 12.0f, 12.0f, FALSE, nullptr, mat(MATERIAL_ID_LAMBERTIAN0), this));
		objects.push_back(new SimpleObjectRect(XZ_RECT, Vec3(0.0f, 5.0f, 0.0f), Vec3(0.0f, 0.2f, 0.0f), 3.0f, 3.0f, TRUE, nullptr, mat(MATERIAL_ID_LIGHTSOURCE_WHITE), this));
		{   // a sphere with its own centre, rotated, then translated
			Built *o = new Built();
			add(o, new TranslatedInstance(new RotatedInstance(new SphereHitable(Vec3(1.0f, 2.0f, 3.0f), 0.8f), 30.0f, 1), Vec3(0.25f, -0.5f, 0.125f)), Vec3(0.25f, -0.5f, 0.125f), MATERIAL_ID_METAL);
		}
		{   // a combo of two spheres
			Built *o = new Built();
			add(o, o->combo({ new SphereHitable(Vec3(-1.5f, 0.2f, 0.3f), 0.6f), new SphereHitable(Vec3(-0.9f, 0.5f, 0.1f), 0.5f) }), Vec3(-1.2f, 0.3f, 0.2f), MATERIAL_ID_LAMBERTIAN1);
		}
		{   // rotations out of the factories' order, a translation in the middle
			Built *o = new Built();
			add(o, new RotatedInstance(new TranslatedInstance(new RotatedInstance(new AxisAlignedRectHitable(0, 2, -0.8f, 0.8f, -0.6f, 0.6f, 0.1f, FALSE), 0.7f, 2), Vec3(0.4f, -1.0f, 0.9f)), -0.5f, 0),
			    Vec3(0.4f, -1.0f, 0.9f), MATERIAL_ID_LAMBERTIAN2);
		}
		{   // nested combos under two instances: a glass ball, a small plate, and a pair of balls
			Built *o = new Built();
			IHitable *pair = o->combo({ new SphereHitable(Vec3(0.5f, 0.0f, 0.0f), 0.25f), new TranslatedInstance(new SphereHitable(Vec3(0.0f, 0.0f, 0.0f), 0.2f), Vec3(-0.5f, 0.1f, 0.0f)) });
			IHitable *all = o->combo({ new SphereHitable(Vec3(0.0f, 0.0f, 0.0f), 0.4f), new TranslatedInstance(new AxisAlignedRectHitable(0, 1, -0.3f, 0.3f, -0.3f, 0.3f, 0.0f, FALSE), Vec3(0.0f, 0.6f, 0.0f)), pair });
			add(o, new TranslatedInstance(new RotatedInstance(all, 1.1f, 1), Vec3(2.2f, -1.0f, 0.5f)), Vec3(2.2f, -1.0f, 0.5f), MATERIAL_ID_DIELECTRIC);
		}
		m_lightSources = new LightSources(this, objects, Vec3(0.6f, 0.7f, 0.9f));
		camera->Initialize(Vec3(0.5f, 1.0f, 9.0f), Vec3(0.0f, 0.0f, 0.0f), 45.0f, 1.0f, 10000.0f, 0.05f, 10.0f, 1.0f);
	} else {
		throw std::invalid_argument("World::ConstructWorld: unknown world id");
	}
	m_objectsCount = objects.size();
	m_objects = objects;
	m_objectBVHTree = new SimpleObjectBVHNode(objects);
}
void World::DeconstructWorld()
{
	++m_generation;                                   // tracers that flattened this world must do it again
	if (m_queryCtx) { mirt_destroy(m_queryCtx); m_queryCtx = nullptr; }
	delete m_objectBVHTree; m_objectBVHTree = nullptr; m_objects.clear();
	delete m_lightSources; m_lightSources = nullptr;
	if (m_resources) { m_resources->Unload(); delete m_resources; m_resources = nullptr; }
}
void World::Flatten(FlatScene &out) const
{
	if (!m_objectBVHTree || !m_lightSources) throw std::runtime_error("World::Flatten: world not constructed");
	for (const Object *o : m_objects) o->Flatten(out);   // object (and primitive) order = construction order, as in the reference's list
	out.root = m_objectBVHTree->Flatten(out);
	Vec3 amb = m_lightSources->GetAmbientLight();
	for (int i = 0; i < 3; ++i) out.ambient[i] = amb[i];
}

BOOL World::ClosestHit(const Ray &r, float tMin, float tMax, Vec3 &position) const
{
	auto check = [&](int rc, const char *what) { if (rc != MIRT_OK) throw std::runtime_error(std::string(what) + ": " + mirt_last_error(m_queryCtx)); };
	if (!m_queryCtx) {
		if (MIRT_CHECK_ABI() != MIRT_OK) throw std::runtime_error(std::string("ABI mismatch: ") + mirt_last_error(nullptr));
		check(mirt_create(m_queryDevice, &m_queryCtx), "mirt_create");
		FlatScene flat;
		Flatten(flat);
		mirt_scene view = flat.View();
		check(mirt_upload_scene(m_queryCtx, &view), "mirt_upload_scene");
	}
	const float org[3] = { r.m_org[0], r.m_org[1], r.m_org[2] }, dir[3] = { r.m_dir[0], r.m_dir[1], r.m_dir[2] };
	mirt_hit hit;
	check(mirt_world_hit_batch(m_queryCtx, org, dir, 1, tMin, tMax, 0u, &hit), "mirt_world_hit_batch");
	if (hit.prim < 0) return FALSE;
	position = Vec3(hit.pos[0], hit.pos[1], hit.pos[2]);
	return TRUE;
}

// =========================================================================================== InputListener
void InputListener::RegisterKey(UINT8 k) { m_keyStatus.emplace(k, KeyState{}); }
BOOL InputListener::WhenPressKey(UINT8 k) { auto it = m_keyStatus.find(k); return it == m_keyStatus.end() ? FALSE : it->second.press; }
BOOL InputListener::WhenReleaseKey(UINT8 k) { auto it = m_keyStatus.find(k); return it == m_keyStatus.end() ? FALSE : it->second.release; }
BOOL InputListener::WhenHoldKey(UINT8 k) { auto it = m_keyStatus.find(k); return it == m_keyStatus.end() ? FALSE : it->second.hold; }
void InputListener::NotifyKeyDown(UINT8 k)                               // InputListener.cpp:48-59
{
	auto it = m_keyStatus.find(k); if (it == m_keyStatus.end()) return;
	KeyState &s = it->second;
	s.press = !s.down; s.hold = s.down; s.release = FALSE; s.down = TRUE;
}
void InputListener::NotifyKeyUp(UINT8 k)                                 // InputListener.cpp:61-73
{
	auto it = m_keyStatus.find(k); if (it == m_keyStatus.end()) return;
	KeyState &s = it->second;
	s.press = FALSE; s.hold = FALSE; s.release = s.down; s.down = FALSE;
}
void InputListener::Clear() { for (auto &kv : m_keyStatus) { kv.second.press = FALSE; kv.second.release = FALSE; } }

// =========================================================================================== SimpleCamera
SimpleCamera::SimpleCamera(const World *world, InputListener *inputListener, float aspectRatio)
	: m_aspectRatio(aspectRatio), m_world(world), m_inputListener(inputListener)
{
	if (m_inputListener) for (UINT8 k : { (UINT8)'H', (UINT8)'W', (UINT8)'A', (UINT8)'S', (UINT8)'D', (UINT8)VK_LEFT, (UINT8)VK_UP, (UINT8)VK_RIGHT, (UINT8)VK_DOWN, (UINT8)VK_ESCAPE }) m_inputListener->RegisterKey(k);
}
void SimpleCamera::Initialize(const Vec3 &lookFrom, const Vec3 &lookAt, float fov, float minZ, float maxZ, float aperture, float unitsPerSecond, float radiansPerSecond)
{                                                                         // SimpleCamera.cpp:32-53
	m_initialOrigin = lookFrom; m_initialFocus = lookAt; m_vup = Vec3(0.0f, 1.0f, 0.0f);
	m_fov = fov; m_nearPlane = minZ; m_farPlane = maxZ; m_lensRadius = aperture / 2.0f;
	Reset();
	m_moveSpeed = unitsPerSecond; m_turnSpeed = radiansPerSecond;
}
void SimpleCamera::Reset()                                               // SimpleCamera.cpp:139-174
{
	m_origin = m_initialOrigin; m_focus = m_initialFocus;
	Vec3 look = m_focus - m_origin;
	float len = look.length();
	if (len != 0.0f) {
		m_pitch = -asinf(look.y() / len);
		if (look.z() > 0.0f) m_yaw = atanf(look.x() / look.z());
		else if (look.z() < 0.0f) m_yaw = atanf(look.x() / look.z()) + (float)M_PI;
		else if (look.x() > 0.0f) m_yaw = (float)M_PI / 2.0f;
		else if (look.x() < 0.0f) m_yaw = -(float)M_PI / 2.0f;
	} else {
		m_focus = m_origin + Vec3(0.0f, 0.0f, -1.0f); m_pitch = 0.0f; m_yaw = (float)M_PI;
	}
	InternalUpdate();
}
void SimpleCamera::InternalUpdate()                                      // SimpleCamera.cpp:177-191
{
	float focusDist = (m_origin - m_focus).length();
	m_w = normalize(m_origin - m_focus);
	m_u = normalize(cross(m_vup, m_w));
	m_v = cross(m_w, m_u);
	const float theta = m_fov * (float)M_PI / 180.0f;
	const float halfH = tanf(theta / 2.0f) * focusDist;
	const float halfW = halfH * m_aspectRatio;
	m_viewTopLeftCorner = m_origin - halfW * m_u + halfH * m_v - focusDist * m_w;
	m_viewHorizontal = 2.0f * halfW * m_u;
	m_viewVertical = -2.0f * halfH * m_v;
}
void SimpleCamera::OnUpdate(float elapsedSeconds)                        // SimpleCamera.cpp:62-124
{
	if (!m_inputListener) return;
	if (m_inputListener->WhenReleaseKey(VK_ESCAPE)) { Reset(); return; }
	const BOOL holdA = m_inputListener->WhenHoldKey('A'), holdD = m_inputListener->WhenHoldKey('D');
	const BOOL holdW = m_inputListener->WhenHoldKey('W'), holdS = m_inputListener->WhenHoldKey('S');
	const BOOL holdLeft = m_inputListener->WhenHoldKey(VK_LEFT), holdRight = m_inputListener->WhenHoldKey(VK_RIGHT);
	const BOOL holdUp = m_inputListener->WhenHoldKey(VK_UP), holdDown = m_inputListener->WhenHoldKey(VK_DOWN);
	const BOOL holdAnyKey = holdA || holdD || holdW || holdS || holdLeft || holdRight || holdUp || holdDown;
	Vec3 move(0.0f, 0.0f, 0.0f);
	if (holdA) move += Vec3(-1.0f, 0.0f, 0.0f);
	if (holdD) move += Vec3(1.0f, 0.0f, 0.0f);
	if (holdW) move += Vec3(0.0f, 0.0f, -1.0f);
	if (holdS) move += Vec3(0.0f, 0.0f, 1.0f);
	if (holdLeft) m_yaw += m_turnSpeed * elapsedSeconds;
	if (holdRight) m_yaw -= m_turnSpeed * elapsedSeconds;
	if (holdUp) m_pitch -= m_turnSpeed * elapsedSeconds;
	if (holdDown) m_pitch += m_turnSpeed * elapsedSeconds;
	if (holdAnyKey) {
		move.normalize();
		move *= m_moveSpeed * elapsedSeconds;
		Vec3 offset = m_u * move.x() + m_w * move.z();
		m_origin += offset;
		float r = cosf(m_pitch);
		Vec3 lookDir(r * sinf(m_yaw), -sinf(m_pitch), r * cosf(m_yaw));
		m_focus = m_origin + lookDir;
		AutoFocus(lookDir);
		InternalUpdate();
	}
	if (m_inputListener->WhenReleaseKey('H')) HelpInfo();
}
BOOL SimpleCamera::AutoFocus(const Vec3 &lookDir)                        // SimpleCamera.cpp:202-214
{
	Vec3 pos;
	if (m_world && m_world->ClosestHit(Ray(m_origin, lookDir), 0.01f, FLT_MAX, pos)) m_focus = pos;
	return TRUE;
}
void SimpleCamera::HelpInfo()
{
	cout << "[SimpleCamera] " << m_origin << " > " << m_focus << " fov " << m_fov << " aperture " << m_lensRadius * 2.0f << endl;
}
void SimpleCamera::Export(mirt_camera &c) const
{
	for (int i = 0; i < 3; ++i) {
		c.origin[i] = m_origin[i]; c.top_left[i] = m_viewTopLeftCorner[i]; c.horizontal[i] = m_viewHorizontal[i];
		c.vertical[i] = m_viewVertical[i]; c.u[i] = m_u[i]; c.v[i] = m_v[i];
	}
	c.lens_radius = m_lensRadius;
}

// =========================================================================================== OutputImage / PPM
OutputImage::OutputImage(UINT32 width, UINT32 height, const char *name) : m_width(width), m_height(height), m_name(name)   // OutputImage.cpp:8-19
{
	m_aspectRatio = static_cast<float>(width) / static_cast<float>(height);
	m_dataSizeInByte = (UINT64)m_width * m_height * m_pixelSizeInByte;
	m_data = new UINT8[m_dataSizeInByte ? m_dataSizeInByte : 1]();
}
OutputImage::~OutputImage() { delete[] m_data; m_data = nullptr; }
static inline UINT8 to_byte(float c) { float s = c * 255.0f; return (s == s) ? (UINT8)(static_cast<UINT32>(s) & 0xFF) : 0; }
void OutputImage::RenderAsRainbow()                                      // OutputImage.cpp:31-49
{
	for (UINT32 j = 0; j < m_height; j++) for (UINT32 i = 0; i < m_width; i++) {
		UINT8 *p = m_data + ((size_t)j * m_width + i) * 4;
		p[0] = to_byte(float(i) / float(m_width - 1)); p[1] = to_byte(1.0f - float(j) / float(m_height - 1)); p[2] = to_byte(0.2f); p[3] = 255;
	}
	m_isDirty = TRUE;
}
void OutputImage::RenderAsRed()
{
	for (size_t i = 0; i < (size_t)m_width * m_height; i++) { UINT8 *p = m_data + i * 4; p[0] = 255; p[1] = 0; p[2] = 0; p[3] = 255; }
	m_isDirty = TRUE;
}
void OutputImage::Render(const Vec3 *pixels, UINT32 pixelCount)           // OutputImage.cpp:62-88
{
	if (pixelCount && pixelCount != m_width * m_height) throw std::invalid_argument("OutputImage::Render: unmatching image size and data size");
	for (size_t i = 0; i < (size_t)m_width * m_height; ++i) {
		Vec3 col = pixels[i];
		col.clamp(Vec3(0.0f, 0.0f, 0.0f), Vec3(1.0f, 1.0f, 1.0f));
		UINT8 *p = m_data + i * 4;
		p[0] = to_byte(col.r()); p[1] = to_byte(col.g()); p[2] = to_byte(col.b()); p[3] = 255;
	}
	m_isDirty = TRUE;
}
void OutputImage::Output()                                               // OutputImage.cpp:255-259
{
	std::string path = m_outputDir + "/" + m_name + ".ppm";
	PPMImageMaker::OutputRGBA8ToFile(path.c_str(), m_width, m_height, m_data, m_dataSizeInByte);
}
void PPMImageMaker::OutputRGBA8ToFile(const char *fileName, UINT32 w, UINT32 h, const UINT8 *rgba8, UINT64 dataSizeInByte)
{
	if (!rgba8) throw std::invalid_argument("PPMImageMaker: invalid data ptr");
	if (dataSizeInByte && dataSizeInByte != (UINT64)w * h * 4) throw std::invalid_argument("PPMImageMaker: unmatching image size and data size");
	cout << "[PPMImageMaker] Save output image to PPM file: " << fileName << " ..." << endl;
	if (mirt_write_ppm(fileName, w, h, rgba8) != MIRT_OK) throw std::runtime_error(std::string("PPMImageMaker: cannot write ") + fileName);
	cout << "[PPMImageMaker] Done" << endl;
}
void OutputImage::OutputBinary()
{
	std::string path = m_outputDir + "/" + m_name + ".ppm";
	PPMImageMaker::OutputRGBA8ToBinaryFile(path.c_str(), m_width, m_height, m_data, m_dataSizeInByte);
}
void PPMImageMaker::OutputRGBA8ToBinaryFile(const char *fileName, UINT32 w, UINT32 h, const UINT8 *rgba8, UINT64 dataSizeInByte)
{
	if (!rgba8) throw std::invalid_argument("PPMImageMaker: invalid data ptr");
	if (dataSizeInByte && dataSizeInByte != (UINT64)w * h * 4) throw std::invalid_argument("PPMImageMaker: unmatching image size and data size");
	if (mirt_write_image(fileName, MIRT_IMG_P6, w, h, rgba8, nullptr) != MIRT_OK) throw std::runtime_error(std::string("PPMImageMaker: cannot write ") + fileName);
}

// =========================================================================================== HomemadeRayTracer
// ABI guard (include/mirt.h): this layer's struct sizes against the loaded libmirt.so, once at load
static const bool g_abi_ok = MIRT_CHECK_ABI() == MIRT_OK;
HomemadeRayTracer::HomemadeRayTracer(InputListener *inputListener, OutputImage *, const World *world) : m_inputListener(inputListener), m_world(world) {}
HomemadeRayTracer::~HomemadeRayTracer() { if (m_ctx) mirt_destroy(m_ctx); m_ctx = nullptr; ReleaseMulti(); }
void HomemadeRayTracer::ReleaseMulti() { if (m_multi) mirt_multi_destroy(m_multi); m_multi = nullptr; }
void HomemadeRayTracer::SetDevices(const std::vector<int> &ids)
{
	if (ids.size() > MIRT_MULTI_MAX_DEVICES) throw std::invalid_argument("HomemadeRayTracer::SetDevices: too many devices");
	ReleaseMulti();
	m_devices = ids.size() > 1 ? ids : std::vector<int>();
	if (ids.size() == 1) SetDevice(ids[0]);
}
void HomemadeRayTracer::SetDeviceCount(int n)
{
	if (n < 1 || n > mirt_device_count()) throw std::invalid_argument("HomemadeRayTracer::SetDeviceCount: the node has " + std::to_string(mirt_device_count()) + " GPU(s)");
	std::vector<int> ids(n);
	for (int i = 0; i < n; ++i) ids[i] = i;
	SetDevices(ids);
}
void HomemadeRayTracer::OnInit()                                          // HomemadeRayTracer.cpp:36-43
{
	cout << "[HomemadeRayTracer] Init" << endl;
	if (m_inputListener) for (UINT8 k : { (UINT8)VK_SPACE, (UINT8)'H', (UINT8)'N', (UINT8)'M' }) m_inputListener->RegisterKey(k);
}
void HomemadeRayTracer::OnUpdate(const SimpleCamera *camera, OutputImage *image)   // HomemadeRayTracer.cpp:45-70
{
	if (!m_inputListener) return;
	if (m_inputListener->WhenReleaseKey(VK_SPACE)) TraceRay(camera, image);
	if (m_inputListener->WhenReleaseKey('H')) HelpInfo();
	if (m_inputListener->WhenReleaseKey('N')) { m_enableNormalDisplay = !m_enableNormalDisplay; cout << "[HomemadeRayTracer][NormalDisplay] " << (m_enableNormalDisplay ? "Enabled" : "Disabled") << endl; }
	if (m_inputListener->WhenReleaseKey('M')) { m_enable1SPP = !m_enable1SPP; cout << "[HomemadeRayTracer][1SPP] " << (m_enable1SPP ? "Enabled" : "Disabled") << endl; }
}
void HomemadeRayTracer::OnDestroy() { cout << "[HomemadeRayTracer] Destroy" << endl; if (m_ctx) mirt_destroy(m_ctx); m_ctx = nullptr; m_uploadedWorld = nullptr; ReleaseMulti(); }
void HomemadeRayTracer::HelpInfo()
{
	cout << "=============HomemadeRayTracer (MI355X)============" << endl
	     << "  [space] render  [n] normal display  [m] 1-SPP  [h] help" << endl
	     << "[NormalDisplay] " << (m_enableNormalDisplay ? "Enabled" : "Disabled") << "  [1SPP] " << (m_enable1SPP ? "Enabled" : "Disabled")
	     << "  [SPP] " << m_spp << "  [depth] " << m_maxDepth << endl;
}
void HomemadeRayTracer::Prepare(const SimpleCamera *camera)
{
	auto check = [&](int rc, const char *what) { if (rc != MIRT_OK) throw std::runtime_error(std::string(what) + ": " + mirt_last_error(m_ctx)); };
	if (!g_abi_ok) throw std::runtime_error(std::string("libmirt_host.so and libmirt.so were built from different revisions of include/mirt.h: ") + mirt_last_error(nullptr));
	const bool fresh = !m_ctx;
	if (fresh) check(mirt_create(m_device, &m_ctx), "mirt_create");
	// The reference reads the live pointer graph on every TraceRay; here the flattened copy on the device is kept until the
	// world says it changed (ConstructWorld / DeconstructWorld / Adopt / Touch bump its generation).
	if (fresh || m_uploadedWorld != m_world || m_uploadedGeneration != m_world->Generation()) {
		FlatScene flat;
		m_world->Flatten(flat);
		mirt_scene view = flat.View();
		check(mirt_upload_scene(m_ctx, &view), "mirt_upload_scene");
		m_uploadedWorld = m_world; m_uploadedGeneration = m_world->Generation();
	}
	mirt_camera cam; camera->Export(cam);
	check(mirt_set_camera(m_ctx, &cam), "mirt_set_camera");
}
void HomemadeRayTracer::TraceRayProgressive(const SimpleCamera *camera, OutputImage *image, UINT32 firstSample, UINT32 sampleCount)
{
	if (!camera || !image || !m_world) throw std::invalid_argument("HomemadeRayTracer::TraceRayProgressive: null argument");
	Prepare(camera);
	mirt_render_params p{};
	p.struct_size = (uint32_t)sizeof p; m_stats.struct_size = (uint32_t)sizeof m_stats;
	p.width = image->m_width; p.height = image->m_height; p.spp_begin = firstSample; p.spp_count = sampleCount; p.max_depth = m_maxDepth; p.seed = m_seed;
	p.flags = MIRT_FLAG_ACCEL_AUTO | (m_enableNormalDisplay ? MIRT_FLAG_NORMAL_DISPLAY : 0u);
	if (mirt_render_progressive(m_ctx, &p, nullptr, nullptr, image->m_data, &m_stats) != MIRT_OK)
		throw std::runtime_error(std::string("mirt_render_progressive: ") + mirt_last_error(m_ctx));
	image->m_isDirty = TRUE;
}
void HomemadeRayTracer::TraceRay(const SimpleCamera *camera, OutputImage *image)   // HomemadeRayTracer.cpp:90-154
{
	if (!camera || !image || !m_world) throw std::invalid_argument("HomemadeRayTracer::TraceRay: null argument");
	cout << "[HomemadeRayTracer] TraceRay ..." << endl;
	cout << "[HomemadeRayTracer] SPP: " << (m_enable1SPP ? 1u : m_spp) << endl;
	auto check = [&](int rc, const char *what) { if (rc != MIRT_OK) throw std::runtime_error(std::string(what) + ": " + mirt_last_error(m_ctx)); };
	mirt_render_params p{};
	p.struct_size = (uint32_t)sizeof p; m_stats.struct_size = (uint32_t)sizeof m_stats;
	p.width = image->m_width; p.height = image->m_height; p.spp_begin = 0; p.spp_count = m_spp; p.max_depth = m_maxDepth; p.seed = m_seed;
	p.flags = MIRT_FLAG_ACCEL_AUTO | (m_enable1SPP ? MIRT_FLAG_ONE_SPP : 0u) | (m_enableNormalDisplay ? MIRT_FLAG_NORMAL_DISPLAY : 0u);
	if (m_devices.size() > 1 && !m_enable1SPP) {
		// the samples of every pixel split over the GPUs of the node, summed onto m_devices[0] (mirt_multi)
		auto mcheck = [&](int rc, const char *what) { if (rc != MIRT_OK) throw std::runtime_error(std::string(what) + ": " + mirt_multi_last_error(m_multi)); };
		if (!g_abi_ok) throw std::runtime_error(std::string("ABI mismatch between libmirt_host.so and libmirt.so: ") + mirt_last_error(nullptr));
		const bool fresh = !m_multi;
		if (fresh) mcheck(mirt_multi_create(m_devices.data(), (int)m_devices.size(), &m_multi), "mirt_multi_create");
		if (fresh || m_multiWorld != m_world || m_multiGeneration != m_world->Generation()) {
			FlatScene flat;
			m_world->Flatten(flat);
			mirt_scene view = flat.View();
			mcheck(mirt_multi_upload_scene(m_multi, &view), "mirt_multi_upload_scene");
			m_multiWorld = m_world; m_multiGeneration = m_world->Generation();
		}
		mirt_camera cam; camera->Export(cam);
		mcheck(mirt_multi_set_camera(m_multi, &cam), "mirt_multi_set_camera");
		m_multiInfo.struct_size = (uint32_t)sizeof m_multiInfo;
		mcheck(mirt_multi_render(m_multi, &p, nullptr, nullptr, image->m_data, &m_stats, &m_multiInfo), "mirt_multi_render");
	} else {
		Prepare(camera);
		check(mirt_render(m_ctx, &p, nullptr, nullptr, image->m_data, &m_stats), "mirt_render");   // device resolve == OutputImage::Render
	}
	image->m_isDirty = TRUE;
	cout << "[HomemadeRayTracer] Done (" << m_stats.rays << " rays, " << m_stats.device_ms << " ms on the GPU)" << endl;
}

// =========================================================================================== C shim for harnesses
// Lets a ctypes harness build a reference world through the C++ API and read back the flattened
// arrays (tests compare them with the Python mirror and the oracle's tree).
struct mirt_host_world { World world; SimpleCamera *cam; InputListener input; FlatScene flat; mirt_scene view; mirt_camera mcam; std::string err; };
extern "C" {
mirt_host_world *mirt_host_world_create(int world_id, float aspect, uint64_t seed, char *err, size_t errlen)
{
	try {
		auto *h = new mirt_host_world();
		Randomizer::Seed(seed);
		h->cam = new SimpleCamera(&h->world, &h->input, aspect);
		h->world.ConstructWorld((WorldID)world_id, h->cam);
		h->world.Flatten(h->flat);
		h->view = h->flat.View();
		h->cam->Export(h->mcam);
		return h;
	} catch (const std::exception &e) { if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return nullptr; }
}
const mirt_scene *mirt_host_world_scene(const mirt_host_world *h) { return &h->view; }
const mirt_camera *mirt_host_world_camera(const mirt_host_world *h) { return &h->mcam; }
void mirt_host_world_destroy(mirt_host_world *h) { if (!h) return; h->world.DeconstructWorld(); delete h->cam; delete h; }
// what the flattener still refuses: a leaf under more than MIRT_MAX_CHAIN_OPS instances.  Builds Translate^n(Sphere) and flattens it;
// returns 0 when it flattened (n_prims, n_ops and whether the primitive took the chain form go out), -1 with the message when refused.
int mirt_host_try_chain(uint32_t n_translates, float cx, uint32_t *n_prims, uint32_t *n_ops, uint32_t *chain_form, char *err, size_t errlen)
{
	try {
		IHitable *h = new SphereHitable(Vec3(cx, 0.0f, 0.0f), 0.5f);
		for (uint32_t k = 0; k < n_translates; ++k) h = new TranslatedInstance(h, Vec3(0.125f, 0.0f, 0.0f));
		Lambertian m(nullptr);
		SimpleTexture2D_SingleColor grey(Vec3(0.5f, 0.5f, 0.5f));
		m.m_albedo = &grey;
		Object o; o.m_hitable = h; o.m_material = &m; h->BindMaterial(&m);
		FlatScene flat;
		o.Flatten(flat);
		*n_prims = (uint32_t)flat.prim_kind.size(); *n_ops = (uint32_t)flat.op_kind.size(); *chain_form = flat.prim_chain[0] != 0;
		return 0;
	} catch (const std::exception &e) { if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
// whole reference flow: ConstructWorld -> TraceRay -> image bytes (and optionally <dir>/<name>.ppm)
// scripted fly-through of the camera: frame f holds the (up to 4) key codes keys[4*f..4*f+3] (0 = none) for dt
// seconds, then the camera is exported.  AutoFocus runs on the GPU.
int mirt_host_camera_script(int world_id, float aspect, uint64_t scene_seed, const uint8_t *keys, uint32_t n_frames, float dt,
                            mirt_camera *out_per_frame, char *err, size_t errlen)
{
	try {
		InputListener input; World world;
		Randomizer::Seed(scene_seed);
		SimpleCamera camera(&world, &input, aspect);
		world.ConstructWorld((WorldID)world_id, &camera);
		for (uint32_t f = 0; f < n_frames; ++f) {
			for (int k = 0; k < 4; ++k) if (keys[4 * f + k]) { input.NotifyKeyDown(keys[4 * f + k]); input.NotifyKeyDown(keys[4 * f + k]); }   // press, then the auto-repeat that means "hold" (InputListener.cpp:48-59)
			camera.OnUpdate(dt);
			for (int k = 0; k < 4; ++k) if (keys[4 * f + k]) input.NotifyKeyUp(keys[4 * f + k]);
			input.Clear();
			camera.Export(out_per_frame[f]);
		}
		world.DeconstructWorld();
		return 0;
	} catch (const std::exception &e) { if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
// DecodeTGA through the C shim: returns 0 and the size; rgba (capacity cap bytes) receives R,G,B,A, row 0 = top
int mirt_host_decode_tga(const char *path, uint32_t *w, uint32_t *h, uint8_t *rgba, size_t cap, char *err, size_t errlen)
{
	try {
		SimpleTexture2D_TGAImage img(path);
		*w = img.m_width; *h = img.m_height;
		if (img.m_pixelData.size() > cap) throw std::runtime_error("output buffer too small");
		memcpy(rgba, img.m_pixelData.data(), img.m_pixelData.size());
		return 0;
	} catch (const std::exception &e) { if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
int mirt_host_render(int world_id, uint32_t w, uint32_t h, uint32_t spp, uint32_t depth, uint64_t scene_seed, uint64_t seed, int one_spp, int normal,
                     uint8_t *rgba8_out, const char *ppm_dir, char *err, size_t errlen)
{
	try {
		InputListener input; OutputImage image(w, h, "OutputImage"); World world;
		Randomizer::Seed(scene_seed);
		SimpleCamera camera(&world, &input, image.m_aspectRatio);
		world.ConstructWorld((WorldID)world_id, &camera);
		HomemadeRayTracer rt(&input, &image, &world);
		rt.OnInit();
		rt.SetSamplesPerPixel(spp); rt.SetMaxDepth(depth); rt.SetSeed(seed); rt.Enable1SPP(one_spp); rt.EnableNormalDisplay(normal);
		rt.TraceRay(&camera, &image);
		if (rgba8_out) memcpy(rgba8_out, image.m_data, image.m_dataSizeInByte);
		if (ppm_dir) { image.m_outputDir = ppm_dir; image.Output(); }
		rt.OnDestroy();
		world.DeconstructWorld();
		return 0;
	} catch (const std::exception &e) { if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
// One World object and one tracer, two scenes in a row: ConstructWorld(a), TraceRay, DeconstructWorld, ConstructWorld(b),
// TraceRay.  The second image must be world b's (the tracer's flattened copy follows World::Generation()).
int mirt_host_render_rebuilt(int world_a, int world_b, uint32_t w, uint32_t h, uint32_t spp, uint64_t scene_seed, uint64_t seed,
                             uint8_t *rgba8_first, uint8_t *rgba8_second, char *err, size_t errlen)
{
	try {
		InputListener input; OutputImage image(w, h, "OutputImage"); World world;
		SimpleCamera camera(&world, &input, image.m_aspectRatio);
		HomemadeRayTracer rt(&input, &image, &world);
		rt.OnInit();
		rt.SetSamplesPerPixel(spp); rt.SetSeed(seed); rt.Enable1SPP(FALSE);
		const int ids[2] = { world_a, world_b };
		uint8_t *outs[2] = { rgba8_first, rgba8_second };
		for (int k = 0; k < 2; ++k) {
			if (k) world.DeconstructWorld();
			Randomizer::Seed(scene_seed);
			world.ConstructWorld((WorldID)ids[k], &camera);
			rt.TraceRay(&camera, &image);
			if (outs[k]) memcpy(outs[k], image.m_data, image.m_dataSizeInByte);
		}
		rt.OnDestroy();
		world.DeconstructWorld();
		return 0;
	} catch (const std::exception &e) { if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
// TraceRay over several devices (HomemadeRayTracer::SetDevices): image bytes + what mirt_multi reports
int mirt_host_render_devices(int world_id, uint32_t w, uint32_t h, uint32_t spp, uint64_t scene_seed, uint64_t seed, const int *devices, int n_devices,
                             uint8_t *rgba8_out, mirt_multi_info *info_out, uint64_t *rays_out, char *err, size_t errlen)
{
	try {
		InputListener input; OutputImage image(w, h, "OutputImage"); World world;
		Randomizer::Seed(scene_seed);
		SimpleCamera camera(&world, &input, image.m_aspectRatio);
		world.ConstructWorld((WorldID)world_id, &camera);
		HomemadeRayTracer rt(&input, &image, &world);
		rt.OnInit();
		rt.SetSamplesPerPixel(spp); rt.SetSeed(seed); rt.Enable1SPP(FALSE);
		rt.SetDevices(std::vector<int>(devices, devices + n_devices));
		rt.TraceRay(&camera, &image);
		if (rgba8_out) memcpy(rgba8_out, image.m_data, image.m_dataSizeInByte);
		if (info_out) *info_out = rt.LastMultiInfo();
		if (rays_out) *rays_out = rt.LastStats().rays;
		rt.OnDestroy();
		world.DeconstructWorld();
		return 0;
	} catch (const std::exception &e) { if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; } return -1; }
}
}
