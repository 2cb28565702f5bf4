//! ref_golden.rs — a dumper a maintainer of KristinnVikarJ/nbody-simulation can run ONCE, where `cargo` exists, to PIN the
//! CPU oracle of this repository (oracle/nbody_oracle.hpp) to the reference's own arithmetic.  It cannot be compiled in the
//! build image of this repository (no Rust toolchain, no network): it is shipped as text, like INTEGRATION.md's binding.
//!
//! How to use (in a checkout of the reference at the commit this repository was surveyed against):
//!   1. copy this file to `src/ref_golden.rs`;
//!   2. in `src/main.rs` add `mod ref_golden;` under `mod bvh_tree;` (line 1) and, as the FIRST statements of `fn main()`
//!      (line 81, before the event loop and the window are created):
//!          if let Some(k) = std::env::args().position(|a| a == "--golden") {
//!              let a: Vec<String> = std::env::args().collect();
//!              ref_golden::run(&a[k + 1], &a[k + 2]);
//!              return Ok(());
//!          }
//!   3. `cargo run --release -- --golden <this repo>/tests/golden/reference_inputs <this repo>/tests/golden/from_reference`
//!   4. in this repository: `python -m pytest tests/test_from_reference.py` — it compares the oracle with every dumped array
//!      BIT FOR BIT (tree of the first build, its permutation, the first force map, the rows after every dumped step).
//!
//! Nothing here changes the reference's arithmetic: it calls `BVHTree::from`, `BVHTree::calculate_gravity`,
//! `World::bvh_sum_gravity` and `World::update` as they are (src/bvh_tree.rs:56-158, src/main.rs:348-425).  The inputs are read
//! from files (the reference's own `World::new` draws from unseeded generators, src/main.rs:276-346, so it cannot produce a
//! reproducible scene); a child module may read its ancestors' private items, so no visibility has to change.
//!
//! Files, all raw little-endian, per case directory `<in>/<case>/`: `pos0.f32` [n][2], `vel0.f32` [n][2], `weight.u32` [n],
//! `steps.txt` (ascending step numbers to dump, whitespace separated).  Written to `<out>/<case>/`:
//!   bvh_is_leaf.i32 [m], bvh_mass.u32 [m], bvh_count.i64 [m] (particles under the node), bvh_geom.f32 [m][6]
//!     = offset.x offset.y size.x size.y cog.x cog.y — the nodes of the FIRST build in pre-order (node, children[0], children[1]);
//!   perm_pos.f32 [n][2], perm_weight.u32 [n] — the particle array as `BVHTree::from` left it (its in-place partition);
//!   acc0.f32 [n][2] — `bvh_sum_gravity` of every row of the snapshot taken BEFORE the build (main.rs:398, :406-416), THETA as compiled;
//!   step_<k>_pos.f32, step_<k>_vel.f32 [n][2], step_<k>_weight.u32 [n] — `world.particles` after k calls of `World::update`.
use std::fs;
use std::io::Write;
use std::path::Path;

use pathfinder_geometry::vector::vec2f;

use crate::bvh_tree::BVHTree;
use crate::{Counting, Particle, Vec2, World, STEP_SIZE};

fn read_f32(p: &Path) -> Vec<f32> {
    fs::read(p).unwrap().chunks_exact(4).map(|b| f32::from_le_bytes([b[0], b[1], b[2], b[3]])).collect()
}
fn read_u32(p: &Path) -> Vec<u32> {
    fs::read(p).unwrap().chunks_exact(4).map(|b| u32::from_le_bytes([b[0], b[1], b[2], b[3]])).collect()
}
fn write_bytes(p: &Path, bytes: &[u8]) {
    fs::File::create(p).unwrap().write_all(bytes).unwrap();
}
fn write_f32(p: &Path, v: &[f32]) {
    write_bytes(p, &v.iter().flat_map(|x| x.to_le_bytes()).collect::<Vec<u8>>());
}
fn write_u32(p: &Path, v: &[u32]) {
    write_bytes(p, &v.iter().flat_map(|x| x.to_le_bytes()).collect::<Vec<u8>>());
}
fn write_i32(p: &Path, v: &[i32]) {
    write_bytes(p, &v.iter().flat_map(|x| x.to_le_bytes()).collect::<Vec<u8>>());
}
fn write_i64(p: &Path, v: &[i64]) {
    write_bytes(p, &v.iter().flat_map(|x| x.to_le_bytes()).collect::<Vec<u8>>());
}

struct Flat {
    is_leaf: Vec<i32>,
    mass: Vec<u32>,
    count: Vec<i64>,
    geom: Vec<f32>,
}

// pre-order: the node, then children[0], then children[1]; returns the number of particles under the node
fn flatten(t: &BVHTree, out: &mut Flat) -> i64 {
    let at = out.is_leaf.len();
    let cog: Vec2 = t.get_center_of_gravity(); // leaf: unweighted mean (NaN for an empty leaf); root: as calculate_gravity left it
    let mass: u32 = t.get_total_mass();
    match t {
        BVHTree::Leaf { children, boundary } => {
            out.is_leaf.push(1);
            out.mass.push(mass);
            out.count.push(children.len() as i64);
            out.geom.extend_from_slice(&[boundary.offset.x(), boundary.offset.y(), boundary.size.x(), boundary.size.y(), cog.x(), cog.y()]);
            children.len() as i64
        }
        BVHTree::Root { boundary, children, .. } => {
            out.is_leaf.push(0);
            out.mass.push(mass);
            out.count.push(0);
            out.geom.extend_from_slice(&[boundary.offset.x(), boundary.offset.y(), boundary.size.x(), boundary.size.y(), cog.x(), cog.y()]);
            let n = flatten(&children[0], out) + flatten(&children[1], out);
            out.count[at] = n;
            n
        }
    }
}

fn positions(ps: &[Particle]) -> Vec<f32> {
    ps.iter().flat_map(|p| [p.position.x(), p.position.y()]).collect()
}
fn velocities(ps: &[Particle]) -> Vec<f32> {
    ps.iter().flat_map(|p| [p.velocity.x(), p.velocity.y()]).collect()
}
fn weights(ps: &[Particle]) -> Vec<u32> {
    ps.iter().map(|p| p.weight).collect()
}

fn run_case(dir: &Path, out: &Path) {
    let pos = read_f32(&dir.join("pos0.f32"));
    let vel = read_f32(&dir.join("vel0.f32"));
    let w = read_u32(&dir.join("weight.u32"));
    let steps: Vec<u32> = fs::read_to_string(dir.join("steps.txt")).unwrap().split_whitespace().map(|s| s.parse().unwrap()).collect();
    let n = w.len();
    assert!(pos.len() == 2 * n && vel.len() == 2 * n);
    let particles: Vec<Particle> = (0..n)
        .map(|i| Particle { position: vec2f(pos[2 * i], pos[2 * i + 1]), velocity: vec2f(vel[2 * i], vel[2 * i + 1]), weight: w[i] })
        .collect();
    fs::create_dir_all(out).unwrap();

    // the first build, on a copy: tree, permutation, force map of the snapshot (exactly the statements of main.rs:398-416)
    {
        let cloned = particles.clone();
        let mut permuted = particles.clone();
        let mut tree = BVHTree::from(permuted.as_mut());
        tree.calculate_gravity();
        let mut flat = Flat { is_leaf: vec![], mass: vec![], count: vec![], geom: vec![] };
        flatten(&tree, &mut flat);
        write_i32(&out.join("bvh_is_leaf.i32"), &flat.is_leaf);
        write_u32(&out.join("bvh_mass.u32"), &flat.mass);
        write_i64(&out.join("bvh_count.i64"), &flat.count);
        write_f32(&out.join("bvh_geom.f32"), &flat.geom);
        let mut acc: Vec<f32> = Vec::with_capacity(2 * n);
        for particle in cloned.iter() {
            let mut a = vec2f(0.0, 0.0);
            World::bvh_sum_gravity(&particle.clone().into(), &tree, &mut a);
            acc.push(a.x());
            acc.push(a.y());
        }
        write_f32(&out.join("acc0.f32"), &acc);
        drop(tree);
        write_f32(&out.join("perm_pos.f32"), &positions(&permuted));
        write_u32(&out.join("perm_weight.u32"), &weights(&permuted));
    }

    // whole steps: World::update as it is (main.rs:388-425), STEP_SIZE as compiled
    let mut world = World { particles };
    let mut counter = Counting { build_bvh: 0.0, sum_gravity: 0.0, post_calculations: 0.0 };
    let mut done = 0u32;
    for &k in steps.iter() {
        while done < k {
            world.update(STEP_SIZE, &mut counter);
            done += 1;
        }
        write_f32(&out.join(format!("step_{}_pos.f32", k)), &positions(&world.particles));
        write_f32(&out.join(format!("step_{}_vel.f32", k)), &velocities(&world.particles));
        write_u32(&out.join(format!("step_{}_weight.u32", k)), &weights(&world.particles));
    }
    println!("{}: n {} steps {:?} -> {}", dir.display(), n, steps, out.display());
}

pub fn run(in_dir: &str, out_dir: &str) {
    let mut cases: Vec<_> = fs::read_dir(in_dir).unwrap().map(|e| e.unwrap().path()).filter(|p| p.is_dir()).collect();
    cases.sort();
    for c in cases {
        run_case(&c, &Path::new(out_dir).join(c.file_name().unwrap()));
    }
}
