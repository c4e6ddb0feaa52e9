// dump_gltf.cpp -- test infrastructure, BUILD CONTAINER ONLY: prints what the reference's own parser, TinyGLTF (the
// library GLTFSceneImporter::ImportScenesFromFile calls, /root/reference/src/core/GLTFSceneImporter.cpp:20-72), makes of a
// glTF file, so that tests/test_scene_ref_cpu.py can hold nebulae_amd.scene.load_gltf against it.  Links against
// /root/reference/vendor/TinyGLTF/tiny_gltf.cc compiled where it lies (oracle/ref_tinygltf/Makefile); nothing of the
// reference is copied, and nothing built here travels to the GPU box's tests (they skip when oracle/_ref is absent).
//
// usage: dump_gltf <file.gltf|file.glb> <out.bin> [--missing-buffers-as-zeros]   -> JSON on stdout, raw payloads in out.bin
//   per primitive: the POSITION / NORMAL / TEXCOORD_0 / TANGENT accessors de-strided to tight float arrays and the indices
//   widened to uint32 -- the values the importer's byte streams + strides address (GLTFSceneImporter.cpp:476-625);
//   per node: the local matrix TinyGLTF hands out (matrix, or T/R/S), children; per material the factors and texture -> image
//   indices the importer reads; per image: width, height and the RGBA8 pixels as decoded by stb_image (TinyGLTF's default
//   loader: the bytes GLTFSceneImporter::SubmitD3D12Images uploads as R8G8B8A8_UNORM, :143-160).
// --missing-buffers-as-zeros: a .bin that is stripped from the checkout (assets/sponza/Sponza.bin) is read as zeros of the
//   length the JSON declares, so that the structure of the file can still be compared (parse-only; accessor payloads are
//   then not dumped).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "json.hpp" // (nlohmann::json, vendored beside tiny_gltf.h in the reference checkout)
#include "tiny_gltf.h"

static bool g_zeros = false;
static std::vector<std::pair<std::string, size_t>> g_declared; // uri -> byteLength, from a first pass over the JSON

static bool FileExistsOrDeclared(const std::string& path, void* ud)
{
    if (tinygltf::FileExists(path, ud))
        return true;
    if (!g_zeros)
        return false;
    for (auto& d : g_declared)
        if (path.size() >= d.first.size() && path.compare(path.size() - d.first.size(), d.first.size(), d.first) == 0)
            return true;
    return false;
}

static bool ReadWholeFileOrZeros(std::vector<unsigned char>* out, std::string* err, const std::string& path, void* ud)
{
    if (tinygltf::FileExists(path, ud))
        return tinygltf::ReadWholeFile(out, err, path, ud);
    for (auto& d : g_declared)
        if (path.size() >= d.first.size() && path.compare(path.size() - d.first.size(), d.first.size(), d.first) == 0) {
            out->assign(d.second, 0);
            return true;
        }
    return false;
}

static void put_floats(std::ofstream& blob, const std::vector<float>& v) { blob.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * 4)); }

int main(int argc, char** argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: dump_gltf <file> <out.bin> [--missing-buffers-as-zeros]\n");
        return 2;
    }
    const std::string path = argv[1];
    g_zeros = argc > 3 && !strcmp(argv[3], "--missing-buffers-as-zeros");
    tinygltf::TinyGLTF loader;
    tinygltf::Model model;
    std::string err, warn;
    if (g_zeros) { // which external buffers does the file declare?  (plain JSON read, no TinyGLTF involved)
        std::ifstream f(path);
        nlohmann::json j = nlohmann::json::parse(f, nullptr, false);
        if (j.is_object() && j.contains("buffers"))
            for (auto& b : j["buffers"])
                if (b.contains("uri") && b.contains("byteLength"))
                    g_declared.emplace_back(b["uri"].get<std::string>(), b["byteLength"].get<size_t>());
        tinygltf::FsCallbacks fs = {&FileExistsOrDeclared, &tinygltf::ExpandFilePath, &ReadWholeFileOrZeros, &tinygltf::WriteWholeFile, nullptr};
        loader.SetFsCallbacks(fs);
    }
    const bool glb = path.size() > 4 && path.substr(path.size() - 4) == ".glb";
    const bool ok = glb ? loader.LoadBinaryFromFile(&model, &err, &warn, path) : loader.LoadASCIIFromFile(&model, &err, &warn, path);
    if (!ok) {
        fprintf(stderr, "TinyGLTF failed: %s\n", err.c_str());
        return 1;
    }
    std::ofstream blob(argv[2], std::ios::binary);
    size_t off = 0;
    nlohmann::json out;
    out["warn"] = warn;
    out["default_scene"] = model.defaultScene;
    for (auto& s : model.scenes)
        out["scenes"].push_back(s.nodes);
    for (auto& n : model.nodes) {
        nlohmann::json jn;
        jn["mesh"] = n.mesh;
        jn["children"] = n.children;
        jn["matrix"] = n.matrix;
        jn["translation"] = n.translation;
        jn["rotation"] = n.rotation;
        jn["scale"] = n.scale;
        out["nodes"].push_back(jn);
    }
    auto tight = [&](int accessor_index, std::vector<float>& dst, int& comps) {
        const tinygltf::Accessor& a = model.accessors[accessor_index];
        const tinygltf::BufferView& bv = model.bufferViews[a.bufferView];
        const tinygltf::Buffer& buf = model.buffers[bv.buffer];
        comps = tinygltf::GetNumComponentsInType(a.type);
        const int stride = a.ByteStride(bv);
        dst.resize(a.count * comps);
        const unsigned char* base = buf.data.data() + bv.byteOffset + a.byteOffset;
        for (size_t i = 0; i < a.count; ++i)
            memcpy(&dst[i * comps], base + i * stride, comps * 4);
    };
    for (auto& m : model.meshes) {
        nlohmann::json jm;
        for (auto& p : m.primitives) {
            nlohmann::json jp;
            jp["material"] = p.material;
            jp["mode"] = p.mode;
            for (const char* name : {"POSITION", "NORMAL", "TEXCOORD_0", "TANGENT"}) {
                auto it = p.attributes.find(name);
                if (it == p.attributes.end())
                    continue;
                const tinygltf::Accessor& a = model.accessors[it->second];
                nlohmann::json ja;
                ja["count"] = a.count;
                ja["component_type"] = a.componentType;
                ja["type"] = a.type;
                ja["stride"] = a.ByteStride(model.bufferViews[a.bufferView]);
                ja["min"] = a.minValues;
                ja["max"] = a.maxValues;
                if (!g_zeros && a.componentType == TINYGLTF_COMPONENT_TYPE_FLOAT) {
                    std::vector<float> v;
                    int comps = 0;
                    tight(it->second, v, comps);
                    ja["offset"] = off;
                    ja["floats"] = v.size();
                    put_floats(blob, v);
                    off += v.size() * 4;
                }
                jp["attributes"][name] = ja;
            }
            if (p.indices >= 0) {
                const tinygltf::Accessor& a = model.accessors[p.indices];
                const tinygltf::BufferView& bv = model.bufferViews[a.bufferView];
                nlohmann::json ji;
                ji["count"] = a.count;
                ji["component_type"] = a.componentType;
                ji["stride"] = a.ByteStride(bv);
                if (!g_zeros) {
                    const unsigned char* base = model.buffers[bv.buffer].data.data() + bv.byteOffset + a.byteOffset;
                    const int stride = a.ByteStride(bv);
                    std::vector<uint32_t> idx(a.count);
                    for (size_t i = 0; i < a.count; ++i) {
                        const unsigned char* q = base + i * stride;
                        idx[i] = a.componentType == TINYGLTF_COMPONENT_TYPE_UNSIGNED_INT     ? *reinterpret_cast<const uint32_t*>(q)
                                 : a.componentType == TINYGLTF_COMPONENT_TYPE_UNSIGNED_SHORT ? *reinterpret_cast<const uint16_t*>(q)
                                                                                             : *q;
                    }
                    ji["offset"] = off;
                    blob.write(reinterpret_cast<const char*>(idx.data()), (std::streamsize)(idx.size() * 4));
                    off += idx.size() * 4;
                }
                jp["indices"] = ji;
            }
            jm.push_back(jp);
        }
        out["meshes"].push_back(jm);
    }
    for (auto& m : model.materials) {
        nlohmann::json jm;
        const auto& pbr = m.pbrMetallicRoughness;
        jm["base_color_factor"] = pbr.baseColorFactor;
        jm["metallic_factor"] = pbr.metallicFactor;
        jm["roughness_factor"] = pbr.roughnessFactor;
        auto image_of = [&](int tex) { return tex >= 0 ? model.textures[tex].source : -1; };
        jm["base_color_image"] = image_of(pbr.baseColorTexture.index);
        jm["metallic_roughness_image"] = image_of(pbr.metallicRoughnessTexture.index);
        jm["normal_image"] = image_of(m.normalTexture.index);
        jm["alpha_mode"] = m.alphaMode;
        jm["double_sided"] = m.doubleSided;
        out["materials"].push_back(jm);
    }
    for (auto& im : model.images) {
        nlohmann::json ji;
        ji["uri"] = im.uri;
        ji["width"] = im.width;
        ji["height"] = im.height;
        ji["component"] = im.component;
        ji["bits"] = im.bits;
        ji["offset"] = off;
        ji["bytes"] = im.image.size();
        blob.write(reinterpret_cast<const char*>(im.image.data()), (std::streamsize)im.image.size());
        off += im.image.size();
        out["images"].push_back(ji);
    }
    printf("%s\n", out.dump().c_str());
    return 0;
}
