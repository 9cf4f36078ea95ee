// trunk_bounds.hip - micro-benchmarks that PRICE the sample-per-XCD persistent trunk proposed in VERDICT r02 item 1
// before building it (not part of libafldm_hip.so; built by tools/proto/run_trunk_bounds.py):
//   1. k_stream: the rate at which ONE XCD can stream a weight matrix that all 8 XCDs read at the same time
//      (mode 1: every XCD reads the whole buffer, its 32 workgroups a 1/32 slice each) against the chip-wide
//      rate of the present kernels (mode 0: 256 workgroups, 1/256 slice each, the buffer is read once).
//   2. k_xcd_barrier: a per-XCD counter barrier + 4 KB hand-over between the workgroups of one XCD, with the
//      counter updated by L2-scope (workgroup-scope encoding: no sc1) or agent-scope atomics and the payload read
//      by plain or sc1 (L1-bypassing) loads; every word is checked, so the cheap forms are also shown (in)valid.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}

// ---------------------------------------------------------------------------------------------- 1. streaming
extern "C" __global__ __launch_bounds__(512) void k_stream(const u32x4* __restrict__ w, size_t n16, int mode, int passes,
                                                          unsigned* sink) {
  const int groups = mode == 0 ? gridDim.x : gridDim.x / 8;
  const int slice = mode == 0 ? blockIdx.x : blockIdx.x / 8;       // (block b lands on XCD b % 8: speed only)
  const size_t per = n16 / groups;
  const u32x4* src = w + (size_t)slice * per;
  u32x4 acc = {0, 0, 0, 0};
  for (int ps = 0; ps < passes; ++ps) {
    for (size_t i = threadIdx.x; i + 7 * 512 < per; i += 8 * 512) {
      u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[i + k * 512];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = acc[0];
}

// ---------------------------------------------------------------------------------------------- 2. XCD-local barrier
// state: ticket[8] (rank hand-out), arrive[8] (monotonic per-XCD arrival counters), all[1] (chip-wide start rendezvous),
//        nx[8] (workgroups per XCD, written once by rank 0 after the rendezvous), err[1]
struct BarState {
  unsigned ticket[8];
  unsigned arrive[8 * 32];     // one 128-byte line per XCD
  unsigned all;
  unsigned pad[31];
  unsigned err;
  unsigned stale_words;
};

template <int ATOM_AGENT, int LOAD_SC1>
__device__ __forceinline__ void run_rounds(BarState* s, unsigned* buf, int rounds, int work_words) {
  __shared__ int sh[4];
  const int tid = threadIdx.x;
  if (tid == 0) {
    const int x = xcc_id();
    const unsigned r = __hip_atomic_fetch_add(&s->ticket[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&s->all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&s->all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(2);
    sh[0] = x;
    sh[1] = (int)r;
    sh[2] = (int)__hip_atomic_load(&s->ticket[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // workgroups on this XCD
    sh[3] = spins >= (1u << 24);
  }
  __syncthreads();
  const int x = sh[0], rank = sh[1], n = sh[2];
  if (sh[3]) {
    if (tid == 0) atomicAdd(&s->err, 1u);
    return;
  }
  unsigned* mine = buf + ((size_t)x * 64 + rank) * work_words;              // this workgroup's record
  const unsigned* next = buf + ((size_t)x * 64 + (rank + 1) % n) * work_words;    // the record it consumes
  unsigned* cnt = &s->arrive[x * 32];
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    // uneven load: a rank-dependent amount of extra work before publishing
    if ((rank + r) % 5 == 0) __builtin_amdgcn_s_sleep(20);
    for (int i = tid; i < work_words; i += blockDim.x) mine[i] = (unsigned)(r * 1000003 + rank * 4099 + i);     // plain stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (ATOM_AGENT) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // no sc1: performed in this XCD's L2
      }
      const unsigned want = (unsigned)(r + 1) * (unsigned)n;
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {      // sc1 load: L1 bypassed
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) {
          atomicAdd(&s->err, 1u);
          break;
        }
      }
      if (ATOM_AGENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    for (int i = tid; i < work_words; i += blockDim.x) {
      unsigned v;
      if (LOAD_SC1) v = __hip_atomic_load(next + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v = next[i];
      if (v != (unsigned)(r * 1000003 + ((rank + 1) % n) * 4099 + i)) ++bad;
    }
    // the record is overwritten next round: the reader must have finished -> second barrier phase of the round
    __syncthreads();
    if (tid == 0) {
      unsigned* cnt2 = cnt + 16;
      if (ATOM_AGENT) __hip_atomic_fetch_add(cnt2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(cnt2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned want = (unsigned)(r + 1) * (unsigned)n;
      unsigned spins = 0;
      while (__hip_atomic_load(cnt2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (bad) atomicAdd(&s->stale_words, bad);
}

extern "C" __global__ __launch_bounds__(256) void k_xcd_barrier(BarState* s, unsigned* buf, int rounds, int work_words, int form) {
  switch (form) {
    case 0: run_rounds<0, 1>(s, buf, rounds, work_words); break;    // L2-scope atomics, sc1 payload loads   (the cheap candidate)
    case 1: run_rounds<0, 0>(s, buf, rounds, work_words); break;    // L2-scope atomics, plain payload loads (expected stale: L1)
    case 2: run_rounds<1, 0>(s, buf, rounds, work_words); break;    // agent release / acquire fences, plain loads (the guide's form)
    case 3: run_rounds<1, 1>(s, buf, rounds, work_words); break;
  }
}

extern "C" int tb_stream(const void* w, size_t bytes, int mode, int passes, void* sink, void* stream) {
  k_stream<<<256, 512, 0, (hipStream_t)stream>>>((const u32x4*)w, bytes / 16, mode, passes, (unsigned*)sink);
  return (int)hipGetLastError();
}
extern "C" int tb_barrier(void* state, void* buf, int rounds, int work_words, int form, int wgs, void* stream) {
  k_xcd_barrier<<<wgs, 256, 0, (hipStream_t)stream>>>((BarState*)state, (unsigned*)buf, rounds, work_words, form);
  return (int)hipGetLastError();
}
extern "C" int tb_state_bytes() { return (int)sizeof(BarState); }
