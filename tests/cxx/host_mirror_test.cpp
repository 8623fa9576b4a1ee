// host_mirror_test.cpp -- exercises the C++ mirror of the reference interface (hostcxx/scene.hpp) the
// way app/Main.hs uses the original: compileFor -> initialOutput -> apply the closure -> reseed, at the
// reference's native 800x600 / 15 bounces / mainScene, and checks the result against the CPU oracle
// (test infrastructure) bit for bit.  Built and run by tests/test_host_cxx.py.
#include <cstdio>
#include <cstring>

#include "scene.hpp"
#include "../../oracle/pt_oracle.h"

using namespace Scene;

static bool same(const void *a, const void *b, size_t bytes, const char *what)
{
    if (std::memcmp(a, b, bytes) == 0) return true;
    std::printf("MISMATCH in %s\n", what);
    return false;
}

int main()
{
    try {
        const int W = 800, H = 600;                                  // src/Util.hs:186-188
        Device dev(0, W, H, World::mainScene());
        const CompiledFunction compute = compileFor(dev, Trace::Algorithm::Inline);   // app/Main.hs:154
        const Camera camera = World::initialCamera();
        RenderResult seeds = Util::initialOutput(dev, 0x5EED1234ull);                  // app/Main.hs:155

        // oracle-side copy of the initial state and scene
        RenderResult want = seeds;
        std::vector<ora_sphere> sp; std::vector<ora_plane> pl;
        for (const Sphere &s : World::mainScene().spheres)
            sp.push_back(ora_sphere{{s.position.x, s.position.y, s.position.z}, s.radius,
                                    {s.material.color.x, s.material.color.y, s.material.color.z}, s.material.illuminance,
                                    s.material.brdf.tag, s.material.brdf.parameter});
        for (const Plane &p : World::mainScene().planes)
            pl.push_back(ora_plane{{p.position.x, p.position.y, p.position.z}, {p.direction.x, p.direction.y, p.direction.z},
                                   {p.material.color.x, p.material.color.y, p.material.color.z}, p.material.illuminance,
                                   p.material.brdf.tag, p.material.brdf.parameter});
        const ora_scene scene{sp.data(), (int)sp.size(), pl.data(), (int)pl.size()};
        const ora_camera ocam{{camera.position.x, camera.position.y, camera.position.z},
                              {camera.rotation.x, camera.rotation.y, camera.rotation.z}, camera.fov};
        bool ok = true;
        {   // genSeeds made deterministic: device seeding == oracle seeding
            std::vector<uint32_t> a(W * H), b(W * H), c(W * H), d(W * H);
            ora_gen_seeds(0x5EED1234ull, 0, (int64_t)W * H, a.data(), b.data(), c.data(), d.data());
            ok &= same(a.data(), seeds.sfc_a.data(), a.size() * 4, "seed plane a");
            ok &= same(d.data(), seeds.sfc_counter.data(), d.size() * 4, "seed plane counter");
        }
        // value = compute' initialCamera (0, seeds), then once more   (app/Main.hs:157, :209-211)
        std::pair<int, RenderResult> value{0, seeds};
        value = compute(camera, value);
        value = compute(camera, value);
        ok &= value.first == 2;
        ora_render_inline(&scene, &ocam, W, H, 15, 2, nullptr, nullptr, want.r.data(), want.g.data(), want.b.data(),
                          want.sfc_a.data(), want.sfc_b.data(), want.sfc_c.data(), want.sfc_counter.data(), ora_max_threads());
        ok &= same(want.r.data(), value.second.r.data(), want.r.size() * 4, "r");
        ok &= same(want.g.data(), value.second.g.data(), want.g.size() * 4, "g");
        ok &= same(want.b.data(), value.second.b.data(), want.b.size() * 4, "b");
        ok &= same(want.sfc_a.data(), value.second.sfc_a.data(), want.sfc_a.size() * 4, "sfc a");
        ok &= same(want.sfc_b.data(), value.second.sfc_b.data(), want.sfc_b.size() * 4, "sfc b");
        ok &= same(want.sfc_c.data(), value.second.sfc_c.data(), want.sfc_c.size() * 4, "sfc c");
        ok &= same(want.sfc_counter.data(), value.second.sfc_counter.data(), want.sfc_counter.size() * 4, "sfc counter");
        // reseed keeps the colour (Util.hs:134-135)
        const RenderResult reseeded = Util::reseed(dev, 99, value.second);
        ok &= same(reseeded.r.data(), value.second.r.data(), reseeded.r.size() * 4, "colour after reseed");
        ok &= std::memcmp(reseeded.sfc_a.data(), value.second.sfc_a.data(), reseeded.sfc_a.size() * 4) != 0;
        // ---- the resident flow at the same size (INTEGRATION.md "resident wiring"): computationLoop's batching
        // (single samples up to 100 iterations, then doTimes batchSize, app/Main.hs:208-211), a reseed in between
        // (:231), the read-outs graphicsLoop needs (:346-351) -- against the oracle doing the same sample by sample.
        {
            Resident res(dev, Trace::Algorithm::Inline);
            res.reset(0x5EED1234ull);
            RenderResult ref = seeds;
            auto oracle_samples = [&](int n) {
                ora_render_inline(&scene, &ocam, W, H, 15, n, nullptr, nullptr, ref.r.data(), ref.g.data(), ref.b.data(),
                                  ref.sfc_a.data(), ref.sfc_b.data(), ref.sfc_c.data(), ref.sfc_counter.data(), ora_max_threads());
            };
            res.compute(camera); res.compute(camera); res.compute(camera);          // three single samples
            res.compute(camera, 30);                                                 // doTimes 30
            oracle_samples(33);
            ok &= res.iterations() == 33;
            RenderResult got = res.value();
            ok &= same(ref.r.data(), got.r.data(), ref.r.size() * 4, "resident r after 33 samples");
            ok &= same(ref.sfc_c.data(), got.sfc_c.data(), ref.sfc_c.size() * 4, "resident sfc c after 33 samples");
            res.reseed(4242);                                                        // run <$> reseed acc
            ora_gen_seeds(4242, 0, (int64_t)W * H, ref.sfc_a.data(), ref.sfc_b.data(), ref.sfc_c.data(), ref.sfc_counter.data());
            res.compute(camera, 7);
            oracle_samples(7);
            got = res.value();
            ok &= same(ref.g.data(), got.g.data(), ref.g.size() * 4, "resident g after reseed + 7");
            ok &= same(ref.sfc_a.data(), got.sfc_a.data(), ref.sfc_a.size() * 4, "resident sfc a after reseed + 7");
            std::vector<float> cr, cg, cb, rgb; std::vector<uint8_t> rgba;
            res.colour(cr, cg, cb);
            ok &= same(ref.b.data(), cb.data(), cb.size() * 4, "colour planes for graphicsLoop");
            res.present(&rgb, &rgba);
            bool present_ok = rgb.size() == 3 * ref.r.size() && rgba.size() == 4 * ref.r.size();
            for (size_t i = 0; present_ok && i < ref.r.size(); i += 997) {           // texture.rgb / u_iterations
                present_ok &= rgb[3 * i] == ref.r[i] / 40.0f && rgb[3 * i + 1] == ref.g[i] / 40.0f && rgb[3 * i + 2] == ref.b[i] / 40.0f;
                present_ok &= rgba[4 * i + 3] == 255;
            }
            if (!present_ok) std::printf("MISMATCH in present\n");
            ok &= present_ok;
            // a camera move: run <$> initialOutput again (app/Main.hs:306), iterations start over
            res.reset(7);
            ok &= res.iterations() == 0;
            res.colour(cr, cg, cb);
            for (size_t i = 0; i < cr.size(); i += 1013) ok &= cr[i] == 0.0f;
        }
        // error behaviour: exception with a code, like a Haskell exception out of runN
        try { Device bad(1 << 20); ok = false; } catch (const PtmiError &e) { ok &= e.code == PTMI_ENODEVICE; }
        std::printf(ok ? "host mirror OK (800x600, 15 bounces: closure flow 2 samples, resident flow 40 samples with batching, reseed and present -- bit-identical to the oracle)\n" : "host mirror FAILED\n");
        return ok ? 0 : 1;
    } catch (const PtmiError &e) {
        std::printf("PtmiError %d: %s\n", e.code, e.what());
        return 2;
    }
}
