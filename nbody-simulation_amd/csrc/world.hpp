// C++ host-side mirror of the reference's step interface, above the C ABI (include/nbody_hip.h).
//
//   struct Particle { position, velocity: Vec2, weight: u32 }          /root/reference src/main.rs:193-198
//   struct Counting { build_bvh, sum_gravity, post_calculations }      src/main.rs:74-79
//   struct World { particles: Vec<Particle> }                          src/main.rs:37-39
//   impl World { fn new() -> Self; fn update(&mut self, delta: f32, counter: &mut Counting) }   :276, :388
//
// Same names and argument meaning; `particles` is refreshed from the device on demand (the reference clones the
// whole vector to the render thread after every step it can deliver, main.rs:137-139).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nbody_hip.h"

namespace nbody_host {

struct Vec2 { float x, y; };
struct Particle { Vec2 position, velocity; uint32_t weight; };
using Counting = nbody_counting;

enum class Method { Bvh, Quad, Direct };

class World {
 public:
  std::vector<Particle> particles;

  explicit World(std::vector<Particle> init, Method method = Method::Bvh, int device = 0, const nbody_params* params = nullptr)
      : particles(std::move(init)), method_(method) {
    check(nbody_create(&ctx_, device), "nbody_create");
    upload(params);
  }
  // The same world on several GPUs of one node: `update` below does not change (main.rs:120 stays one call); the library
  // shards each step's targets over `devices` and exchanges the results itself (nbody_create_multi).
  World(std::vector<Particle> init, Method method, const std::vector<int>& devices, const nbody_params* params = nullptr)
      : particles(std::move(init)), method_(method) {
    check(nbody_create_multi(&ctx_, (int)devices.size(), devices.data()), "nbody_create_multi");
    upload(params);
  }
  ~World() { nbody_destroy(ctx_); }
  World(const World&) = delete;
  World& operator=(const World&) = delete;

 private:
  void upload(const nbody_params* params) {
    if (params) check(nbody_set_params(ctx_, params), "nbody_set_params");
    std::vector<float> pos(2 * particles.size()), vel(2 * particles.size());
    std::vector<uint32_t> w(particles.size());
    for (size_t i = 0; i < particles.size(); ++i) {
      pos[2 * i] = particles[i].position.x; pos[2 * i + 1] = particles[i].position.y;
      vel[2 * i] = particles[i].velocity.x; vel[2 * i + 1] = particles[i].velocity.y;
      w[i] = particles[i].weight;
    }
    check(nbody_upload_f32(ctx_, (int64_t)particles.size(), pos.data(), vel.data(), w.data()), "nbody_upload_f32");
  }

 public:
  // World::update, main.rs:388-425
  void update(float delta, Counting& counter) {
    int rc = method_ == Method::Direct ? nbody_update_direct_f32(ctx_, delta, 1, &counter)
                                       : nbody_update_tree_f32(ctx_, method_ == Method::Bvh ? NBODY_TREE_BVH : NBODY_TREE_QUAD,
                                                               delta, 1, &counter);
    check(rc, "update");
    stale_ = true;
  }

  // The same step without waiting for it (nbody_update_tree_async_f32): the loop of main.rs:118-139 — update, then hand a
  // snapshot over if the channel has room — keeps the device busy from one iteration to the next.  `wait` folds the phase
  // times into `counter`.  (The direct path's steps are hundreds of milliseconds: it stays synchronous.)
  void update_async(float delta) {
    if (method_ == Method::Direct) { Counting c{}; update(delta, c); return; }
    check(nbody_update_tree_async_f32(ctx_, method_ == Method::Bvh ? NBODY_TREE_BVH : NBODY_TREE_QUAD, delta, 1), "update_async");
    stale_ = true;
  }
  void wait(Counting& counter) {
    check(nbody_wait(ctx_), "nbody_wait");
    Counting total{};
    check(nbody_get_counting(ctx_, &total), "nbody_get_counting");
    counter = total;
  }

  // Brings `particles` up to date with the device (the reference's hand-off clone, main.rs:138).
  const std::vector<Particle>& snapshot() {
    if (stale_) {
      const size_t n = particles.size();
      std::vector<float> pos(2 * n), vel(2 * n);
      std::vector<uint32_t> w(n);
      check(nbody_download_f32(ctx_, pos.data(), vel.data(), w.data(), nullptr), "nbody_download_f32");
      for (size_t i = 0; i < n; ++i)
        particles[i] = Particle{{pos[2 * i], pos[2 * i + 1]}, {vel[2 * i], vel[2 * i + 1]}, w[i]};
      stale_ = false;
    }
    return particles;
  }

  // The hand-off as a delta stream (the experiment of main.rs:107-134; nbody_delta_* in nbody_hip.h): positions in the
  // order `init` had, as the change against the previous stream.  Returns `updates` at the time of the snapshot.
  uint64_t delta_snapshot(std::vector<uint8_t>& stream) {
    check(nbody_delta_begin(ctx_), "nbody_delta_begin");
    stream.resize(nbody_delta_bound((int64_t)particles.size(), 0));
    size_t bytes = 0;
    uint64_t step = 0;
    check(nbody_delta_end(ctx_, stream.data(), stream.size(), &bytes, &step), "nbody_delta_end");
    stream.resize(bytes);
    return step;
  }

  // Row r of the device image is the particle `init[ids[r]]` (the BVH build permutes the rows every step).
  void rows(std::vector<float>& pos, std::vector<uint32_t>& ids) {
    pos.resize(2 * particles.size());
    ids.resize(particles.size());
    check(nbody_download_f32(ctx_, pos.data(), nullptr, nullptr, ids.data()), "nbody_download_f32");
  }

  // draw(&particles, frame), main.rs:41-72: the render_px x render_px RGBA frame of the current particles.
  void draw(std::vector<uint8_t>& frame, uint32_t height = 100000, uint32_t render_px = 1250) {
    frame.resize((size_t)render_px * render_px * 4);
    check(nbody_render_rgba(ctx_, height, render_px, frame.data()), "nbody_render_rgba");
  }

 private:
  void check(int rc, const char* what) {
    if (rc != NBODY_OK) throw std::runtime_error(std::string(what) + ": " + nbody_last_error(ctx_));
  }
  nbody_ctx* ctx_ = nullptr;
  Method method_;
  bool stale_ = false;
};

}  // namespace nbody_host
