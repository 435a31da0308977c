// Resouces.h (the reference's spelling) -- the material table of Resources::LoadMaterials
// (RayTracer/RayTracer/Resouces.h:3-71, Resouces.cpp:104-209).  Meshes exist only for the rasterizer.
#pragma once
#include "stdafx.h"
#include "Materials.h"      // MaterialUniqueID / MeshUniqueID, the slots of the table

class Mesh;
class IMaterial;
class ITexture2D;

class Resources
{
public:
	void       Load();
	void       Unload();
	Mesh      *GetTheMesh(MeshUniqueID) const { return nullptr; }
	IMaterial *GetTheMaterial(MaterialUniqueID id) const;
	size_t     GetMeshesCount() const { return 0; }
	size_t     GetTexturesCount() const { return m_textures.size(); }
	size_t     GetMaterialsCount() const { return m_materials.size(); }

private:
	void LoadMaterials();
	std::vector<ITexture2D *> m_textures;
	std::vector<IMaterial *>  m_materials;
};
