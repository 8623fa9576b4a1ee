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
// A RenderResult is a VALUE whose planes live on the device until somebody reads them -- what the result of runN is on the
// reference's own GPU backend: compileFor's closure is built on ptmi_render1_chained (the result of one call is found on the
// device by the next; nothing crosses PCIe), r() / g() / b() fetch the colour planes on first use (graphicsLoop,
// app/Main.hs:350), the seed planes come down only if somebody asks.  compileForCopying is ptmi_render1's closure -- seven
// planes each way per call -- kept for comparison.
//
// Header only; link with -lptmi.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
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

class PtmiError : public std::runtime_error {
public:
    PtmiError(int code, const std::string &what) : std::runtime_error(what), code(code) {}
    int code;
};

// type RenderResult = Matrix (Color, SFC32)   (Objects.hs:36): seven row-major planes -- an immutable value.  Either it is a
// STATE of a device context under a token (include/ptmi.h, "the closure, chained"), its planes fetched on first use and kept, or
// it was made from host planes (fromHost: a result that came from elsewhere); copies share one payload, the last one to go
// releases the token.
class RenderResult {
public:
    RenderResult() : p_(std::make_shared<Payload>()) {}
    static RenderResult fromHost(int w, int h, std::vector<float> r, std::vector<float> g, std::vector<float> b, std::vector<uint32_t> sfc_a,
                                 std::vector<uint32_t> sfc_b, std::vector<uint32_t> sfc_c, std::vector<uint32_t> sfc_counter)
    {
        RenderResult out;
        Payload &p = *out.p_;
        p.width = w; p.height = h;
        p.r = std::move(r); p.g = std::move(g); p.b = std::move(b);
        p.a = std::move(sfc_a); p.b2 = std::move(sfc_b); p.c = std::move(sfc_c); p.counter = std::move(sfc_counter);
        p.have_colour = p.have_seeds = true;
        return out;
    }
    static RenderResult onDevice(std::shared_ptr<ptmi_ctx> ctx, uint64_t token, int w, int h)
    {
        RenderResult out;
        out.p_->ctx = std::move(ctx); out.p_->token = token; out.p_->width = w; out.p_->height = h;
        return out;
    }
    int width() const { return p_->width; }
    int height() const { return p_->height; }
    // the token this value is held under by `ctx`, or 0 (it lives on the host only, or in another context)
    uint64_t tokenIn(const ptmi_ctx *ctx) const { return p_->ctx.get() == ctx ? p_->token : 0; }
    bool onHost() const { std::lock_guard<std::mutex> lock(p_->mu); return p_->have_colour && p_->have_seeds; }
    const std::vector<float> &r() const { colour(); return p_->r; }
    const std::vector<float> &g() const { colour(); return p_->g; }
    const std::vector<float> &b() const { colour(); return p_->b; }
    const std::vector<uint32_t> &sfc_a() const { seeds(); return p_->a; }
    const std::vector<uint32_t> &sfc_b() const { seeds(); return p_->b2; }
    const std::vector<uint32_t> &sfc_c() const { seeds(); return p_->c; }
    const std::vector<uint32_t> &sfc_counter() const { seeds(); return p_->counter; }

private:
    struct Payload {
        int width = 0, height = 0;
        std::shared_ptr<ptmi_ctx> ctx;
        uint64_t token = 0;
        std::mutex mu;
        bool have_colour = false, have_seeds = false;
        std::vector<float> r, g, b;
        std::vector<uint32_t> a, b2, c, counter;
        ~Payload() { if (token) ptmi_chain_release(ctx.get(), token); }     // the last copy is gone: the device may reuse the state
    };
    void check(int rc) const { if (rc != PTMI_OK) throw PtmiError(rc, ptmi_last_error(p_->ctx.get())); }
    void colour() const          // A.toVectors texture, the colour part (app/Main.hs:350): three planes come down, once
    {
        Payload &p = *p_;
        std::lock_guard<std::mutex> lock(p.mu);
        if (p.have_colour) return;
        const size_t n = (size_t)p.width * p.height;
        p.r.resize(n); p.g.resize(n); p.b.resize(n);
        check(ptmi_chain_fetch(p.ctx.get(), p.token, p.r.data(), p.g.data(), p.b.data(), nullptr, nullptr, nullptr, nullptr));
        p.have_colour = true;
    }
    void seeds() const
    {
        Payload &p = *p_;
        std::lock_guard<std::mutex> lock(p.mu);
        if (p.have_seeds) return;
        const size_t n = (size_t)p.width * p.height;
        p.a.resize(n); p.b2.resize(n); p.c.resize(n); p.counter.resize(n);
        check(ptmi_chain_fetch(p.ctx.get(), p.token, nullptr, nullptr, nullptr, p.a.data(), p.b2.data(), p.c.data(), p.counter.data()));
        p.have_seeds = true;
    }
    std::shared_ptr<Payload> p_;
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
    const std::shared_ptr<ptmi_ctx> &shared() const { return ctx_; }
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

// the resident planes of the context (ptmi_resize / ptmi_render) as a RenderResult on the host
inline RenderResult download(const Device &dev)
{
    const size_t n = (size_t)dev.width() * dev.height();
    std::vector<float> r(n), g(n), b(n);
    std::vector<uint32_t> sa(n), sb(n), sc(n), sd(n);
    dev.check(ptmi_download_state(dev.get(), r.data(), g.data(), b.data(), sa.data(), sb.data(), sc.data(), sd.data()));
    return RenderResult::fromHost(dev.width(), dev.height(), std::move(r), std::move(g), std::move(b), std::move(sa), std::move(sb), std::move(sc), std::move(sd));
}

// run <$> initialOutput   (Util.hs:204-205; app/Main.hs:155, :306).  genSeeds' OS entropy becomes seed0.  The result stays on the device.
inline RenderResult initialOutput(const Device &dev, uint64_t seed0)
{
    uint64_t token = 0;
    dev.check(ptmi_chain_init_output(dev.get(), dev.width(), dev.height(), seed0, &token));
    return RenderResult::onDevice(dev.shared(), token, dev.width(), dev.height());
}

// run <$> reseed acc      (Util.hs:134-135; app/Main.hs:231): keep colour, replace every RNG state.  An `acc` the device holds is
// reseeded there; any other one has its colour planes uploaded.
inline RenderResult reseed(const Device &dev, uint64_t seed0, const RenderResult &acc)
{
    uint64_t token = 0;
    const uint64_t held = acc.tokenIn(dev.get());
    int rc = held ? ptmi_chain_reseed(dev.get(), seed0, acc.width(), acc.height(), held, 0, nullptr, nullptr, nullptr, &token) : PTMI_ESTALE;
    if (rc == PTMI_ESTALE)
        rc = ptmi_chain_reseed(dev.get(), seed0, acc.width(), acc.height(), 0, 0, acc.r().data(), acc.g().data(), acc.b().data(), &token);
    dev.check(rc);
    return RenderResult::onDevice(dev.shared(), token, acc.width(), acc.height());
}
}  // namespace Util

// ---- compileFor (app/Main.hs:188-191) -----------------------------------------------------------
using Options = Trace::Algorithm;                                                   // app/Main.hs:110
using CompiledFunction =
    std::function<std::pair<int, RenderResult>(const Camera &, const std::pair<int, RenderResult> &)>;   // app/Main.hs:83-84

inline ptmi_camera cameraRecord(const Camera &c) { return ptmi_camera{{c.position.x, c.position.y, c.position.z}, {c.rotation.x, c.rotation.y, c.rotation.z}, c.fov}; }

// let dewit = runN (render config) screenPixels in \c (iterations, acc) -> (iterations + 1, dewit (scalar c) acc)
// on ptmi_render1_chained: an `acc` that is a result of this device is found there (no upload); the new result stays there (no download).
// An `acc` from anywhere else -- host planes, another device's result -- goes through the copy path and gives the same planes.
inline CompiledFunction compileFor(const Device &dev, Options config)
{
    return [dev, config](const Camera &c, const std::pair<int, RenderResult> &state) {
        const RenderResult &acc = state.second;
        const ptmi_camera cam = cameraRecord(c);
        uint64_t token = 0;
        const uint64_t held = acc.tokenIn(dev.get());
        int rc = PTMI_ESTALE;
        if (held)
            rc = ptmi_render1_chained(dev.get(), &cam, (int)config, Trace::maxIterations, acc.width(), acc.height(), held, 0,
                                      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &token,
                                      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (rc == PTMI_ESTALE)                                   // not (or no longer) a state of this device: its planes, from wherever they are
            rc = ptmi_render1_chained(dev.get(), &cam, (int)config, Trace::maxIterations, acc.width(), acc.height(), 0, 0,
                                      acc.r().data(), acc.g().data(), acc.b().data(), acc.sfc_a().data(), acc.sfc_b().data(),
                                      acc.sfc_c().data(), acc.sfc_counter().data(), &token,
                                      nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        dev.check(rc);
        return std::make_pair(state.first + 1, RenderResult::onDevice(dev.shared(), token, acc.width(), acc.height()));
    };
}

// The same closure on ptmi_render1: seven host planes in, seven out, every call (what compileFor was until 0.5; for comparisons).
inline CompiledFunction compileForCopying(const Device &dev, Options config)
{
    return [dev, config](const Camera &c, const std::pair<int, RenderResult> &state) {
        const RenderResult &acc = state.second;
        const size_t n = (size_t)acc.width() * acc.height();
        std::vector<float> r(n), g(n), b(n);
        std::vector<uint32_t> sa(n), sb(n), sc(n), sd(n);
        const ptmi_camera cam = cameraRecord(c);
        dev.check(ptmi_render1(dev.get(), &cam, (int)config, Trace::maxIterations, acc.width(), acc.height(), nullptr, nullptr,
                               acc.r().data(), acc.g().data(), acc.b().data(), acc.sfc_a().data(), acc.sfc_b().data(),
                               acc.sfc_c().data(), acc.sfc_counter().data(),
                               r.data(), g.data(), b.data(), sa.data(), sb.data(), sc.data(), sd.data()));
        return std::make_pair(state.first + 1, RenderResult::fromHost(acc.width(), acc.height(), std::move(r), std::move(g), std::move(b),
                                                                      std::move(sa), std::move(sb), std::move(sc), std::move(sd)));
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
        const ptmi_camera cam = cameraRecord(c);
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
