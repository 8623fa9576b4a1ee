// scene.hpp -- host-side mirror, in C++, of the reference's interface for the hot path.
//
// The reference is compiled Haskell and its toolchain is absent here, so the host layer above the
// C ABI (include/ptmi.h) is written in C++ with the reference's names, argument meaning and error
// behaviour (a failure surfaces as an exception, like one thrown out of runN):
//
//   Scene::Objects   Camera, Sphere, Plane, Material, Brdf, RenderResult   src/Scene/Objects.hs
//   Scene::World     mainScene, initialCamera                              src/Scene/World.hs
//   Scene::Trace     Algorithm { Streams, Inline }                         src/Scene/Trace.hs:68
//   Scene::Util      screenPixels, initialOutput, reseed                   src/Util.hs
//   Scene::compileFor(options) -> CompiledFunction                         app/Main.hs:83-84, :188-191
//
// Header only; link with -lptmi.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ptmi.h"

namespace Scene {

// ---- Scene.Objects -------------------------------------------------------------------------
struct V3 { float x, y, z; };
using Point = V3; using Direction = V3; using Color = V3;          // Objects.hs:40-42

struct Brdf {                                                      // Objects.hs:77-87
    enum Tag : int32_t { Matte = PTMI_MATTE, Glossy = PTMI_GLOSSY } tag;
    float parameter;
};
struct Material { Color color; float illuminance; Brdf brdf; };    // Objects.hs:90-100
struct Sphere { Point position; float radius; Material material; };        // Objects.hs:126-131
struct Plane { Point position; Direction direction; Material material; };  // Objects.hs:103-108
struct Camera { Point position; Direction rotation; int64_t fov; };        // Objects.hs:67-74
struct SceneDescription { std::vector<Sphere> spheres; std::vector<Plane> planes; };   // Objects.hs:60-64

// type RenderResult = Matrix (Color, SFC32)   (Objects.hs:36): seven row-major planes
struct RenderResult {
    int width = 0, height = 0;
    std::vector<float> r, g, b;
    std::vector<uint32_t> sfc_a, sfc_b, sfc_c, sfc_counter;
    void resize(int w, int h)
    {
        width = w; height = h;
        const size_t n = (size_t)w * h;
        r.assign(n, 0); g.assign(n, 0); b.assign(n, 0);
        sfc_a.assign(n, 0); sfc_b.assign(n, 0); sfc_c.assign(n, 0); sfc_counter.assign(n, 0);
    }
};

class PtmiError : public std::runtime_error {
public:
    PtmiError(int code, const std::string &what) : std::runtime_error(what), code(code) {}
    int code;
};

// ---- Scene.World ---------------------------------------------------------------------------
namespace World {
inline Camera initialCamera() { return Camera{{1.0f, -1.6f, -4.8f}, {0.314f, -0.314f, 0.0f}, 90}; }   // World.hs:8-12
inline SceneDescription mainScene()                                                                   // World.hs:15-77
{
    SceneDescription s;
    s.spheres = {
        {{2.0f, 2.0f, -14.0f}, 5.0f, {{1.0f, 0.3f, 0.3f}, 0.0f, {Brdf::Matte, 0.8f}}},
        {{6.0f, 2.0f, -9.0f}, 1.5f, {{0.0f, 0.4f, 0.0f}, 0.0f, {Brdf::Matte, 0.9f}}},
        {{4.5f, 1.0f, -9.0f}, 0.5f, {{0.4f, 0.4f, 1.0f}, 0.0f, {Brdf::Glossy, 1.0f}}},
        {{16.0f, -2.05f, -20.0f}, 0.9f, {{0.8f, 0.8f, 0.8f}, 6942.0f, {Brdf::Glossy, 0.5f}}},
        {{5.0f, 10.0f, 4.0f}, 2.0f, {{0.99f, 0.84f, 0.12f}, 4420.0f, {Brdf::Matte, 1.0f}}},
    };
    s.planes = {
        {{0.0f, -3.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {{0.43f, 0.95f, 0.5f}, 0.0f, {Brdf::Matte, 1.5f}}},
        {{0.0f, 15.0f, 0.0f}, {0.0f, -1.0f, 0.0f}, {{0.26f, 0.68f, 0.88f}, 0.0f, {Brdf::Glossy, 0.9f}}},
    };
    return s;
}
}  // namespace World

// ---- Scene.Trace ---------------------------------------------------------------------------
namespace Trace {
enum class Algorithm { Streams = PTMI_STREAMS, Inline = PTMI_INLINE };   // Trace.hs:68
constexpr int maxIterations = 15;                                        // Trace.hs:80-81, :200
}  // namespace Trace

// ---- the device context (what runN's backend state is to the reference) -------------------------
class Device {
public:
    explicit Device(int device = 0, int screenWidth = 800, int screenHeight = 600,        // Util.hs:186-188
                    const SceneDescription &scene = World::mainScene())
        : width_(screenWidth), height_(screenHeight)
    {
        ptmi_ctx *raw = nullptr;
        const int rc = ptmi_create(&raw, device);
        if (rc != PTMI_OK) throw PtmiError(rc, ptmi_last_error(nullptr));
        ctx_.reset(raw, ptmi_destroy);
        std::vector<ptmi_sphere> sp;
        std::vector<ptmi_plane> pl;
        for (const Sphere &s : scene.spheres)
            sp.push_back(ptmi_sphere{{s.position.x, s.position.y, s.position.z}, s.radius,
                                     {s.material.color.x, s.material.color.y, s.material.color.z},
                                     s.material.illuminance, s.material.brdf.tag, s.material.brdf.parameter});
        for (const Plane &p : scene.planes)
            pl.push_back(ptmi_plane{{p.position.x, p.position.y, p.position.z},
                                    {p.direction.x, p.direction.y, p.direction.z},
                                    {p.material.color.x, p.material.color.y, p.material.color.z},
                                    p.material.illuminance, p.material.brdf.tag, p.material.brdf.parameter});
        check(ptmi_set_scene(raw, sp.data(), (int)sp.size(), pl.data(), (int)pl.size()));
        check(ptmi_resize(raw, width_, height_));
    }
    ptmi_ctx *get() const { return ctx_.get(); }
    static std::string buildId() { return ptmi_build_id(); }      // what the loaded libptmi was built from: print it next to any timing
    int width() const { return width_; }
    int height() const { return height_; }
    void check(int rc) const { if (rc != PTMI_OK) throw PtmiError(rc, ptmi_last_error(ctx_.get())); }

private:
    std::shared_ptr<ptmi_ctx> ctx_;
    int width_, height_;
};

// ---- Util ------------------------------------------------------------------------------------
namespace Util {
// screenPixels :: Matrix (V2 Int), V2 x y at index (Z :. y :. x)   (Util.hs:209-210)
inline std::pair<std::vector<int64_t>, std::vector<int64_t>> screenPixels(int width, int height)
{
    std::vector<int64_t> xs((size_t)width * height), ys((size_t)width * height);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) { xs[(size_t)y * width + x] = x; ys[(size_t)y * width + x] = y; }
    return {xs, ys};
}

inline RenderResult download(const Device &dev)
{
    RenderResult out;
    out.resize(dev.width(), dev.height());
    dev.check(ptmi_download_state(dev.get(), out.r.data(), out.g.data(), out.b.data(), out.sfc_a.data(),
                                  out.sfc_b.data(), out.sfc_c.data(), out.sfc_counter.data()));
    return out;
}

// run <$> initialOutput   (Util.hs:204-205; app/Main.hs:155, :306).  genSeeds' OS entropy becomes seed0.
inline RenderResult initialOutput(const Device &dev, uint64_t seed0)
{
    dev.check(ptmi_init_output(dev.get(), seed0));
    return download(dev);
}

// run <$> reseed acc      (Util.hs:134-135; app/Main.hs:231): keep colour, replace every RNG state
inline RenderResult reseed(const Device &dev, uint64_t seed0, const RenderResult &acc)
{
    dev.check(ptmi_upload_state(dev.get(), acc.r.data(), acc.g.data(), acc.b.data(), nullptr, nullptr, nullptr, nullptr));
    dev.check(ptmi_reseed(dev.get(), seed0));
    return download(dev);
}
}  // namespace Util

// ---- compileFor (app/Main.hs:188-191) -----------------------------------------------------------
using Options = Trace::Algorithm;                                                   // app/Main.hs:110
using CompiledFunction =
    std::function<std::pair<int, RenderResult>(const Camera &, const std::pair<int, RenderResult> &)>;   // app/Main.hs:83-84

// let dewit = runN (render config) screenPixels in \c (iterations, acc) -> (iterations + 1, dewit (scalar c) acc)
inline CompiledFunction compileFor(const Device &dev, Options config)
{
    return [dev, config](const Camera &c, const std::pair<int, RenderResult> &state) {
        const RenderResult &acc = state.second;
        RenderResult out;
        out.resize(acc.width, acc.height);
        const ptmi_camera cam{{c.position.x, c.position.y, c.position.z}, {c.rotation.x, c.rotation.y, c.rotation.z}, c.fov};
        dev.check(ptmi_render1(dev.get(), &cam, (int)config, Trace::maxIterations, acc.width, acc.height, nullptr, nullptr,
                               acc.r.data(), acc.g.data(), acc.b.data(), acc.sfc_a.data(), acc.sfc_b.data(),
                               acc.sfc_c.data(), acc.sfc_counter.data(),
                               out.r.data(), out.g.data(), out.b.data(), out.sfc_a.data(), out.sfc_b.data(),
                               out.sfc_c.data(), out.sfc_counter.data()));
        return std::make_pair(state.first + 1, std::move(out));
    };
}

// ---- the resident flow: the same calls of app/Main.hs with the accumulator left on the device ---------
// computationLoop renders `doTimes batchSize` samples per turn (app/Main.hs:208-211) and graphicsLoop only ever
// reads the three colour planes (app/Main.hs:346-351), so nothing but those has to cross PCIe:
//   Resident r(dev, config);            -- compileFor
//   r.reset(seed0);                     -- run <$> initialOutput          (:155, and :306 on a camera move)
//   r.compute(camera, batchSize);       -- doTimes batchSize (compute camera)   = ONE ptmi_render(n_spp = batchSize)
//   r.reseed(seed0);                    -- run <$> reseed acc             (:231)
//   r.colour() / r.present(rgb, rgba);  -- A.toVectors texture + zipWith3 V3 (:350-351), fs.glsl's divide
class Resident {
public:
    Resident(const Device &dev, Options config) : dev_(dev), config_(config) {}
    int iterations() const { return iterations_; }
    void reset(uint64_t seed0) { dev_.check(ptmi_init_output(dev_.get(), seed0)); iterations_ = 0; }
    void reseed(uint64_t seed0) { dev_.check(ptmi_reseed(dev_.get(), seed0)); }
    // \c (iterations, acc) -> (iterations + n, dewit^n (scalar c) acc); asynchronous
    int compute(const Camera &c, int n = 1)
    {
        const ptmi_camera cam{{c.position.x, c.position.y, c.position.z}, {c.rotation.x, c.rotation.y, c.rotation.z}, c.fov};
        dev_.check(ptmi_render(dev_.get(), &cam, (int)config_, Trace::maxIterations, n));
        return iterations_ += n;
    }
    RenderResult value() const { return Util::download(dev_); }                      // the whole RenderResult (tests, checkpoints)
    void colour(std::vector<float> &r, std::vector<float> &g, std::vector<float> &b) const
    {
        const size_t n = (size_t)dev_.width() * dev_.height();
        r.resize(n); g.resize(n); b.resize(n);
        dev_.check(ptmi_download_color(dev_.get(), r.data(), g.data(), b.data()));
    }
    void present(std::vector<float> *rgb32f, std::vector<uint8_t> *rgba8) const     // texture.rgb / u_iterations (fs.glsl:12)
    {
        const size_t n = (size_t)dev_.width() * dev_.height();
        if (rgb32f) rgb32f->resize(3 * n);
        if (rgba8) rgba8->resize(4 * n);
        dev_.check(ptmi_present(dev_.get(), iterations_, rgb32f ? rgb32f->data() : nullptr, rgba8 ? rgba8->data() : nullptr));
    }

private:
    Device dev_;
    Options config_;
    int iterations_ = 0;
};

}  // namespace Scene
