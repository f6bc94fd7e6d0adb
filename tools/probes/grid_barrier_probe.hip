// Probe (round 5, VERDICT r04 item 1, premise test i): what does a device-wide barrier between the phases of ONE
// persistent launch cost on gfx950, against the dependent kernel boundary it would replace?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/grid_barrier_probe.hip -o tools/probes/bin/grid_barrier_probe
// 256 workgroups of 256 threads, one per CU (the shape of the <= 64x64-map launches of the frame).  Measured:
//   boundary     N dependent launches of a kernel that does the phase's work (or nothing): wall / N
//   flat         one monotonic counter: every workgroup drains its stores, lane 0 = agent release fence + relaxed atomic add,
//                polls the counter with relaxed sc1 loads + s_sleep, then an agent acquire fence      (MI355X_MICROARCH "barrier-counter")
//   xcd          hierarchical: a counter per XCD (workgroups of an XCD share its L2, so only the XCD's last arriver runs the
//                release fence = L2 write-back), that leader arrives on a top counter, waits for all XCDs, then releases its
//                XCD's generation word; every workgroup ends with an agent acquire (L1 invalidate)       ("barrier-xcd")
// Each variant runs (a) empty phases and (b) phases that publish 16 KB per workgroup with plain stores and, after the
// barrier, read and CHECK the 16 KB of another workgroup (every word, consumer L1-warm, every 7th workgroup arriving late):
// a stale read shows up in the error count.  Prints us per barrier / per boundary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WG = 256, NT = 256, WORDS = 4096;     // 16 KB per workgroup and phase
constexpr unsigned SPIN_LIMIT = 1u << 20;           // every spin is bounded: a barrier that cannot complete sets the timeout word

struct Bar {
  unsigned* flat;       // [1]
  unsigned* xcc_cnt;    // [8 * 32] one line each
  unsigned* xcc_gen;    // [8 * 32]
  unsigned* top;        // [1]
  unsigned* census;     // [8 * 32] workgroups per XCD (filled by phase 0 behind a flat barrier)
  unsigned* timeout;    // [1]
  unsigned* errors;     // [1]
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all threads call; `target` = arrivals expected on the flat counter when this barrier completes
__device__ __forceinline__ void barrier_flat(const Bar& b, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(b.flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (ld_relaxed(b.flat) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { *b.timeout = 1; break; } }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// gen = 1, 2, ...: the barrier's ordinal.  xcc = this workgroup's XCD, nx = workgroups on it, nxcd = XCDs in use
__device__ __forceinline__ void barrier_xcd(const Bar& b, unsigned gen, int xcc, unsigned nx, unsigned nxcd) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(b.xcc_cnt + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    if (t == gen * nx - 1) {                     // this XCD's last arriver: publish the XCD's L2, meet the other XCDs, open the gate
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_relaxed(b.top) < gen * nxcd) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { *b.timeout = 1; break; } }
      __hip_atomic_store(b.xcc_gen + xcc * 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (ld_relaxed(b.xcc_gen + xcc * 32) < gen) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { *b.timeout = 1; break; } }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ __forceinline__ void phase_write(unsigned* buf, int wg, int phase) {
  uint4* dst = reinterpret_cast<uint4*>(buf + ((size_t)(phase & 1) * WG + wg) * WORDS);
  for (int i = threadIdx.x; i < WORDS / 4; i += NT) {
    const unsigned v = (unsigned)(phase * 1000003 + wg * 4099 + i * 4);
    dst[i] = make_uint4(v, v + 1, v + 2, v + 3);
  }
}
__device__ __forceinline__ unsigned phase_check(const unsigned* buf, int wg, int phase) {
  const int src = (wg + 37 * (phase + 1)) % WG;
  const uint4* s = reinterpret_cast<const uint4*>(buf + ((size_t)(phase & 1) * WG + src) * WORDS);
  unsigned bad = 0;
  for (int i = threadIdx.x; i < WORDS / 4; i += NT) {
    const unsigned v = (unsigned)(phase * 1000003 + src * 4099 + i * 4);
    const uint4 g = s[i];
    bad += (g.x != v) + (g.y != v + 1) + (g.z != v + 2) + (g.w != v + 3);
  }
  return bad;
}

template <int MODE, bool WORK>      // MODE 0: flat, 1: xcd
__global__ __launch_bounds__(NT) void k_persistent(Bar b, unsigned* buf, int phases) {
  const int wg = blockIdx.x;
  int xcc = 0; unsigned nx = WG, nxcd = 1;
  unsigned flat_n = 0;
  if (MODE == 1) {
    xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;      // HW_REG_XCC_ID (id 20), bits [3:0]
    if (threadIdx.x == 0) __hip_atomic_fetch_add(b.census + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    barrier_flat(b, flat_n += WG);
    nxcd = 0;
    for (int x = 0; x < 8; ++x) { const unsigned c = ld_relaxed(b.census + x * 32); nxcd += c > 0; if (x == xcc) nx = c; }
  }
  unsigned bad = 0;
  for (int p = 0; p < phases; ++p) {
    if (WORK) {
      phase_write(buf, wg, p);
      if (wg % 7 == 0) __builtin_amdgcn_s_sleep(40);      // uneven arrival
    }
    if (MODE == 0) barrier_flat(b, flat_n += WG); else barrier_xcd(b, (unsigned)(p + 1), xcc, nx, nxcd);
    if (WORK) bad += phase_check(buf, wg, p);
  }
  if (WORK && bad) atomicAdd(b.errors, bad);
}

template <bool WORK>
__global__ __launch_bounds__(NT) void k_phase(unsigned* buf, int p, unsigned* errors) {
  if (!WORK) return;
  const int wg = blockIdx.x;
  unsigned bad = p > 0 ? phase_check(buf, wg, p - 1) : 0;
  phase_write(buf, wg, p);
  if (wg % 7 == 0) __builtin_amdgcn_s_sleep(40);        // the same uneven tail as the persistent variant
  if (bad) atomicAdd(errors, bad);
}

int main() {
  Bar b;
  unsigned* pool; CHECK(hipMalloc(&pool, 4096 * sizeof(unsigned)));
  unsigned* buf; CHECK(hipMalloc(&buf, (size_t)2 * WG * WORDS * sizeof(unsigned)));
  auto reset = [&]() {
    CHECK(hipMemset(pool, 0, 4096 * sizeof(unsigned)));
    b.flat = pool; b.top = pool + 64; b.timeout = pool + 128; b.errors = pool + 192;
    b.xcc_cnt = pool + 256; b.xcc_gen = pool + 256 + 8 * 32; b.census = pool + 256 + 16 * 32;
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int P = 400;
  auto report = [&](const char* name, float ms, int n) {
    unsigned host[4096]; CHECK(hipMemcpy(host, pool, sizeof host, hipMemcpyDeviceToHost));
    printf("%-44s %7.2f us each   (%d phases, %.3f ms; errors %u, timeout %u)\n", name, ms * 1e3 / n, n, ms, host[192], host[128]);
  };
  for (int rep = 0; rep < 2; ++rep) {       // second round = warm
    float ms;
    reset(); CHECK(hipEventRecord(e0)); for (int p = 0; p < P; ++p) k_phase<false><<<WG, NT>>>(buf, p, b.errors); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("kernel boundary, empty 256-WG kernels", ms, P);
    reset(); CHECK(hipEventRecord(e0)); for (int p = 0; p < P; ++p) k_phase<true><<<WG, NT>>>(buf, p, b.errors); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("kernel boundary, 16 KB publish + check / WG", ms, P);
    reset(); CHECK(hipEventRecord(e0)); k_persistent<0, false><<<WG, NT>>>(b, buf, P); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("persistent, flat counter barrier, empty", ms, P);
    reset(); CHECK(hipEventRecord(e0)); k_persistent<0, true><<<WG, NT>>>(b, buf, P); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("persistent, flat counter barrier, 16 KB/WG", ms, P);
    reset(); CHECK(hipEventRecord(e0)); k_persistent<1, false><<<WG, NT>>>(b, buf, P); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("persistent, XCD-hierarchical barrier, empty", ms, P);
    reset(); CHECK(hipEventRecord(e0)); k_persistent<1, true><<<WG, NT>>>(b, buf, P); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep) report("persistent, XCD-hierarchical barrier, 16 KB/WG", ms, P);
    if (rep) {
      unsigned host[4096]; CHECK(hipMemcpy(host, pool, sizeof host, hipMemcpyDeviceToHost));
      printf("census (workgroups per XCD):");
      for (int x = 0; x < 8; ++x) printf(" %u", host[256 + 16 * 32 + x * 32]);
      printf("\n");
    }
  }
  return 0;
}
