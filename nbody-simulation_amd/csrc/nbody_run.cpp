// Headless driver: the reference's scene (World::new, /root/reference src/main.rs:276-346, with a seeded
// generator: the reference's own is unseeded) stepped through the C ABI, printing the reference's once-a-second
// block (ups / step / Counting, main.rs:149-156).  The window and the channel of the reference are out of scope; the frame
// its render thread paints (draw, main.rs:41-72) can be written out instead of shown.
//   nbody_run [--gpus G] [--async] [steps=100] [bvh|quad|direct] [seed] [frame_every=0] [frame_prefix=frame] [delta_every=0]
// --gpus G: the same world on devices 0 .. G-1 of this node behind one handle (nbody_create_multi: the step's targets are
// sharded over them, the exchange is an RCCL all-gather inside the library); prints ms/step at the end.
// --async: every step is `update_async` (returns once enqueued), the Counting is fetched with `wait` when it is printed.
// frame_every = k > 0: every k-th step <frame_prefix>_<step>.pam (1250 x 1250 RGBA, Netpbm PAM) is written.
// delta_every = k > 0: every k-th step the positions are taken as a delta stream and the two lines of the commented
// experiment of main.rs:124-133 are printed ("raw: <bytes>" / "comp: <bytes>"); the streams are applied to a decoder and
// its state is compared with the device's at the end.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "world.hpp"

using namespace nbody_host;

static uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static float uni(uint64_t& s) { return (float)((splitmix64(s) >> 40) * (1.0 / 16777216.0)); }

static std::vector<Particle> scene(uint64_t seed) {
  const float TAU = 6.28318530717958647692f;
  std::vector<Particle> p;
  const Vec2 c1{35000.f, 35000.f}, c2{60000.f, 60000.f};
  const float c1lenr2 = 15000000.f;
  p.push_back({c1, {200.f, 250.f}, 75000000u});                       // main.rs:282-286
  p.push_back({c2, {0.f, 0.f}, 750000u});                             // :287-291
  const int m = 100000 / 14 - 1;                                      // :316
  for (int x = 0; x < m; ++x)
    for (int y = 0; y < m; ++y) {
      Vec2 pos{x * 14.f, y * 14.f};
      float dx = pos.x - c2.x, dy = pos.y - c2.y, d2 = dx * dx + dy * dy;
      if (d2 < c1lenr2 && d2 > 500000.f && uni(seed) * ((c1lenr2 - d2) + 1.0f) > 6000000.f) {   // :319-321
        float sc = std::sqrt(std::sqrt(750000.f) / d2);               // :323-324
        p.push_back({pos, {dy * sc, -dx * sc}, 1u});                  // rotate_right :271-273
      }
    }
  for (int i = 0; i < 100000; ++i) {                                  // :334, rand_body :260-269
    float th = uni(seed) * TAU, r = uni(seed);
    float tv = uni(seed) * TAU, rv = uni(seed);
    p.push_back({{std::cos(th) * r * 25000.f + 50000.f, std::sin(th) * r * 25000.f + 50000.f},
                 {std::cos(tv) * rv, std::sin(tv) * rv}, 1u});
  }
  return p;
}

int main(int argc, char** argv) {
  // nbody_run dump <path> [seed]: the generated scene as raw little-endian rows {x, y, vx, vy: f32, weight: u32} — no
  // device involved; tests/test_scene_statistics.py checks it against what World::new (main.rs:276-346) prescribes
  if (argc > 2 && !std::strcmp(argv[1], "dump")) {
    const uint64_t dseed = argc > 3 ? std::strtoull(argv[3], nullptr, 0) : 0xC0FFEEull;
    const std::vector<Particle> p = scene(dseed);
    static_assert(sizeof(Particle) == 20, "Particle is five 4-byte fields");
    FILE* f = std::fopen(argv[2], "wb");
    if (!f || std::fwrite(p.data(), sizeof(Particle), p.size(), f) != p.size()) { std::perror("nbody_run dump"); return 1; }
    std::fclose(f);
    std::printf("len: %zu\n", p.size());
    return 0;
  }
  int gpus = 0;
  if (argc > 2 && !std::strcmp(argv[1], "--gpus")) {
    gpus = std::atoi(argv[2]);
    if (gpus < 1) { std::fprintf(stderr, "nbody_run: --gpus needs a positive count\n"); return 2; }
    argc -= 2;
    argv += 2;
  }
  bool async = false;
  if (argc > 1 && !std::strcmp(argv[1], "--async")) {
    async = true;
    argc -= 1;
    argv += 1;
  }
  int steps = argc > 1 ? std::atoi(argv[1]) : 100;
  Method method = Method::Bvh;
  if (argc > 2 && !std::strcmp(argv[2], "quad")) method = Method::Quad;
  if (argc > 2 && !std::strcmp(argv[2], "direct")) method = Method::Direct;
  uint64_t seed = argc > 3 ? std::strtoull(argv[3], nullptr, 0) : 0xC0FFEEull;
  const int frame_every = argc > 4 ? std::atoi(argv[4]) : 0;
  const char* frame_prefix = argc > 5 ? argv[5] : "frame";
  const int delta_every = argc > 6 ? std::atoi(argv[6]) : 0;
  std::vector<uint8_t> frame, stream;
  nbody_delta_decoder* dec = delta_every > 0 ? nbody_delta_decoder_create() : nullptr;
  try {
    std::vector<int> devices;
    for (int d = 0; d < gpus; ++d) devices.push_back(d);
    World world = gpus > 0 ? World(scene(seed), method, devices) : World(scene(seed), method);
    std::printf("len: %zu\n", world.particles.size());                // main.rs:343
    const auto run_t0 = std::chrono::steady_clock::now();
    Counting counter{};
    auto t0 = std::chrono::steady_clock::now();
    long updates = 0, last = 0;
    for (int s = 0; s < steps; ++s) {
      if (async) world.update_async(0.1f);
      else world.update(0.1f, counter);                               // main.rs:120 (STEP_SIZE)
      ++updates;
      if (frame_every > 0 && updates % frame_every == 0) {
        world.draw(frame);
        char path[512];
        std::snprintf(path, sizeof(path), "%s_%06ld.pam", frame_prefix, updates);
        if (FILE* f = std::fopen(path, "wb")) {
          std::fprintf(f, "P7\nWIDTH 1250\nHEIGHT 1250\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n");
          std::fwrite(frame.data(), 1, frame.size(), f);
          std::fclose(f);
        }
      }
      if (delta_every > 0 && updates % delta_every == 0) {
        world.delta_snapshot(stream);
        std::printf("raw: %zu\ncomp: %zu\n", world.particles.size() * 8, stream.size());
        if (nbody_delta_decoder_apply(dec, stream.data(), stream.size()) != NBODY_OK)
          throw std::runtime_error(nbody_delta_decoder_error(dec));
      }
      auto now = std::chrono::steady_clock::now();
      if (std::chrono::duration<double>(now - t0).count() >= 1.0 || s + 1 == steps) {
        if (async) world.wait(counter);
        std::printf("ups: %ld\nstep: %ld\nCounting { build_bvh: %.6f, sum_gravity: %.6f, post_calculations: %.6f }\n",
                    updates - last, updates, counter.build_bvh, counter.sum_gravity, counter.post_calculations);
        last = updates;
        t0 = now;
      }
    }
    if (dec && nbody_delta_decoder_count(dec) >= 0) {   // what the receiver holds is what the device held at the last stream
      if (steps % delta_every != 0) {
        world.delta_snapshot(stream);
        if (nbody_delta_decoder_apply(dec, stream.data(), stream.size()) != NBODY_OK) throw std::runtime_error(nbody_delta_decoder_error(dec));
      }
      std::vector<float> got(2 * world.particles.size()), pos;
      std::vector<uint32_t> ids;
      nbody_delta_decoder_positions_f32(dec, got.data());
      world.rows(pos, ids);
      size_t bad = 0;
      for (size_t r = 0; r < ids.size(); ++r)
        bad += std::memcmp(&got[2 * (size_t)ids[r]], &pos[2 * r], 8) != 0;
      std::printf("delta streams: receiver state differs from the device in %zu of %zu bodies\n", bad, ids.size());
      if (bad) throw std::runtime_error("delta round trip failed");
    }
    const double run_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - run_t0).count();
    std::printf("gpus: %d\nms/step: %.4f\n", gpus > 0 ? gpus : 1, steps > 0 ? 1e3 * run_s / steps : 0.0);
    const auto& ps = world.snapshot();
    double cx = 0, cy = 0;
    for (auto& q : ps) { cx += q.position.x; cy += q.position.y; }
    std::printf("centroid after %d steps: (%.3f, %.3f)\n", steps, cx / ps.size(), cy / ps.size());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "nbody_run: %s\n", e.what());
    nbody_delta_decoder_destroy(dec);
    return 1;
  }
  nbody_delta_decoder_destroy(dec);
  return 0;
}
