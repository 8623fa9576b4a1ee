// host_mirror_test.cpp -- exercises the C++ mirror of the reference interface (hostcxx/scene.hpp) the
// way app/Main.hs uses the original: compileFor -> initialOutput -> apply the closure -> reseed, at the
// reference's native 800x600 / 15 bounces / mainScene, and checks the result against the CPU oracle
// (test infrastructure) bit for bit.  Built and run by tests/test_host_cxx.py.
#include <chrono>
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

// the oracle's working copy of a RenderResult: seven host planes it renders into
struct Planes7 {
    std::vector<float> r, g, b;
    std::vector<uint32_t> a, b2, c, counter;
    explicit Planes7(const RenderResult &v) : r(v.r()), g(v.g()), b(v.b()), a(v.sfc_a()), b2(v.sfc_b()), c(v.sfc_c()), counter(v.sfc_counter()) {}
    bool equals(const RenderResult &v, const char *what) const
    {
        bool ok = true;
        const size_t bytes = r.size() * 4;
        ok &= same(r.data(), v.r().data(), bytes, what); ok &= same(g.data(), v.g().data(), bytes, what); ok &= same(b.data(), v.b().data(), bytes, what);
        ok &= same(a.data(), v.sfc_a().data(), bytes, what); ok &= same(b2.data(), v.sfc_b().data(), bytes, what);
        ok &= same(c.data(), v.sfc_c().data(), bytes, what); ok &= same(counter.data(), v.sfc_counter().data(), bytes, what);
        return ok;
    }
};

int main()
{
    try {
        const int W = 800, H = 600;                                  // src/Util.hs:186-188
        Device dev(0, W, H, World::mainScene());
        const CompiledFunction compute = compileFor(dev, Trace::Algorithm::Inline);   // app/Main.hs:154
        const Camera camera = World::initialCamera();
        const RenderResult seeds = Util::initialOutput(dev, 0x5EED1234ull);            // app/Main.hs:155

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
        auto oracle_camera = [](const Camera &c) { return ora_camera{{c.position.x, c.position.y, c.position.z}, {c.rotation.x, c.rotation.y, c.rotation.z}, c.fov}; };
        auto oracle_samples = [&](Planes7 &p, const Camera &c, int n) {
            const ora_camera oc = oracle_camera(c);
            ora_render_inline(&scene, &oc, W, H, 15, n, nullptr, nullptr, p.r.data(), p.g.data(), p.b.data(),
                              p.a.data(), p.b2.data(), p.c.data(), p.counter.data(), ora_max_threads());
        };
        bool ok = true;
        {   // genSeeds made deterministic: device seeding == oracle seeding
            std::vector<uint32_t> a(W * H), b(W * H), c(W * H), d(W * H);
            ora_gen_seeds(0x5EED1234ull, 0, (int64_t)W * H, a.data(), b.data(), c.data(), d.data());
            ok &= same(a.data(), seeds.sfc_a().data(), a.size() * 4, "seed plane a");
            ok &= same(d.data(), seeds.sfc_counter().data(), d.size() * 4, "seed plane counter");
        }
        // value = compute' initialCamera (0, seeds), then once more   (app/Main.hs:157, :209-211)
        Planes7 want(seeds);
        std::pair<int, RenderResult> value{0, seeds};
        value = compute(camera, value);
        value = compute(camera, value);
        ok &= value.first == 2;
        ok &= !value.second.onHost();                                // nothing has come down yet
        oracle_samples(want, camera, 2);
        ok &= want.equals(value.second, "closure, 2 samples");
        const Planes7 after2 = want;
        // reseed keeps the colour (Util.hs:134-135)
        const RenderResult reseeded = Util::reseed(dev, 99, value.second);
        ok &= same(reseeded.r().data(), value.second.r().data(), reseeded.r().size() * 4, "colour after reseed");
        ok &= std::memcmp(reseeded.sfc_a().data(), value.second.sfc_a().data(), reseeded.sfc_a().size() * 4) != 0;

        // ---- the closure as computationLoop drives it, with the values left on the device:
        // 100 single-sample calls, a reseed, a camera move (fresh initialOutput, moved camera, three samples) == the oracle;
        // intermediate values die as the loop goes on (their tokens are released by the last copy's destructor).
        {
            std::pair<int, RenderResult> v{0, seeds};
            Planes7 ref(seeds);
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 100; ++k) v = compute(camera, v);
            dev.check(ptmi_synchronize(dev.get()));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100.0;
            std::printf("chained closure: %.1f us per call (800x600, 100 calls, nothing fetched)\n", us);
            oracle_samples(ref, camera, 100);
            ok &= v.first == 100 && ref.equals(v.second, "closure, 100 chained samples");
            ptmi_chain_stats info{};
            dev.check(ptmi_chain_info(dev.get(), &info));
            ok &= info.renders_uploaded == 0 && info.evictions == 0;                // every input was found on the device
            if (info.renders_uploaded != 0 || info.evictions != 0) std::printf("MISMATCH: %llu uploads, %llu evictions in the chain\n", (unsigned long long)info.renders_uploaded, (unsigned long long)info.evictions);
            // run <$> reseed acc (app/Main.hs:231), then a sample
            v.second = Util::reseed(dev, 4242, v.second);
            ora_gen_seeds(4242, 0, (int64_t)W * H, ref.a.data(), ref.b2.data(), ref.c.data(), ref.counter.data());
            v = compute(camera, v);
            oracle_samples(ref, camera, 1);
            ok &= ref.equals(v.second, "closure, reseed + 1 sample");
            // the camera moves: emptyOutput <- initialOutput; compute updatedCamera (0, emptyOutput)   (app/Main.hs:306-319)
            Camera moved = camera;
            moved.position.x += 0.75f; moved.rotation.x += 0.1f;
            std::pair<int, RenderResult> m{0, Util::initialOutput(dev, 7)};
            Planes7 mref(m.second);
            for (int k = 0; k < 3; ++k) m = compute(moved, m);
            oracle_samples(mref, moved, 3);
            ok &= m.first == 3 && mref.equals(m.second, "closure, camera move + 3 samples");
            // an input that is NOT a result of this device: host planes (a result from elsewhere), and another device context's result
            const RenderResult from_host = RenderResult::fromHost(W, H, after2.r, after2.g, after2.b, after2.a, after2.b2, after2.c, after2.counter);
            std::pair<int, RenderResult> h = compute(camera, std::make_pair(2, from_host));
            Planes7 href = after2;
            oracle_samples(href, camera, 1);
            ok &= h.first == 3 && href.equals(h.second, "closure on host planes (copy path)");
            dev.check(ptmi_chain_info(dev.get(), &info));
            ok &= info.renders_uploaded == 1;
            {
                Device other(0, W, H, World::mainScene());
                const RenderResult theirs = compileFor(other, Trace::Algorithm::Inline)(camera, std::make_pair(0, Util::initialOutput(other, 0x5EED1234ull))).second;
                const RenderResult ours = compute(camera, std::make_pair(1, theirs)).second;     // their token means nothing here: the planes travel
                ok &= after2.equals(ours, "closure on another context's result (copy path)");
            }
            // the copying closure (ptmi_render1) gives the same planes
            const RenderResult copied = compileForCopying(dev, Trace::Algorithm::Inline)(camera, std::make_pair(2, from_host)).second;
            ok &= copied.onHost() && href.equals(copied, "compileForCopying");
        }

        // ---- the resident flow at the same size (INTEGRATION.md "resident wiring"): computationLoop's batching
        // (single samples up to 100 iterations, then doTimes batchSize, app/Main.hs:208-211), a reseed in between
        // (:231), the read-outs graphicsLoop needs (:346-351) -- against the oracle doing the same sample by sample.
        {
            Resident res(dev, Trace::Algorithm::Inline);
            res.reset(0x5EED1234ull);
            Planes7 ref(seeds);
            res.compute(camera); res.compute(camera); res.compute(camera);          // three single samples
            res.compute(camera, 30);                                                 // doTimes 30
            oracle_samples(ref, camera, 33);
            ok &= res.iterations() == 33;
            ok &= ref.equals(res.value(), "resident, 33 samples");
            res.reseed(4242);                                                        // run <$> reseed acc
            ora_gen_seeds(4242, 0, (int64_t)W * H, ref.a.data(), ref.b2.data(), ref.c.data(), ref.counter.data());
            res.compute(camera, 7);
            oracle_samples(ref, camera, 7);
            ok &= ref.equals(res.value(), "resident, reseed + 7");
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
        std::printf(ok ? "host mirror OK (800x600, 15 bounces: chained closure 2 + 100 samples with reseed, camera move and foreign inputs, copying closure, "
                         "resident flow 40 samples with batching, reseed and present -- bit-identical to the oracle)\n" : "host mirror FAILED\n");
        return ok ? 0 : 1;
    } catch (const PtmiError &e) {
        std::printf("PtmiError %d: %s\n", e.code, e.what());
        return 2;
    }
}
