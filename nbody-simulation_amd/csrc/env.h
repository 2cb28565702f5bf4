// Environment switches of libnbody_hip.  Internal.
//
// PRODUCT switches — documented in include/nbody_hip.h ("Environment"), read by every build (nbody::env_int):
//   NBODY_TRACE, NBODY_BUILD_THREADS, NBODY_TREE_BUILD_HOST, NBODY_DIRECT_GRAPH, NBODY_DIRECT_NEARFAR, NBODY_STEP_AHEAD,
//   NBODY_WALK_SPLIT, NBODY_MULTI_EXCHANGE, NBODY_MULTI_CHUNKS.
// LAB switches — A/B variants of kernels that lost their comparison, forced slow paths, test hooks, development logs
// (nbody::lab_int / lab_str): honoured ONLY by the laboratory build (`make lab`: -DNBODY_LAB -> lib/libnbody_hip_lab.so, what
// tools/ and the lab-marked tests load).  The product build compiles them to their defaults, and the kernel variants only they
// select are not instantiated in it: a host that links libnbody_hip.so gets one behaviour, whatever NBODY_* it inherits.
#pragma once
#include <cstdlib>

namespace nbody {

inline int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}

#ifdef NBODY_LAB
constexpr bool kLabBuild = true;
inline int lab_int(const char* name, int dflt) { return env_int(name, dflt); }
inline const char* lab_str(const char* name) { return std::getenv(name); }
#else
constexpr bool kLabBuild = false;
constexpr int lab_int(const char*, int dflt) { return dflt; }
constexpr const char* lab_str(const char*) { return nullptr; }
#endif

}  // namespace nbody
