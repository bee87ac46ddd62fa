// l2_handoff.hip — time-boxed probe for DESIGN.md's "layer-specialised CU pipeline with hand-offs through L2":
// does a tensor that one workgroup produces and a workgroup on the SAME XCD consumes a few microseconds later stay out
// of HBM?
//
//   mode 0 (hand-off)  256 workgroups, one per CU.  Blocks b and b+8 share an XCD (round-robin dealing; a wrong guess
//                      is slower, never wrong): they form a producer/consumer pair.  The producer streams NT tiles of
//                      16 KiB (one 32-pixel block of a 256-wide fp16 delta tensor) through a ring of R slots in global
//                      memory; protocol = MI355X_MICROARCH.md "Valid forms", first table row: sc1 (write-through)
//                      payload stores, every storing wave drains vmcnt, workgroup barrier, ONE lane stores the slot's
//                      flag sc1; the consumer polls the flag with an sc1 load (one lane), barrier, sc1 payload loads,
//                      drain, barrier, one lane hands the slot back through a second flag.  Every spin is bounded.
//   mode 1 (baseline)  the same bytes through a buffer that is written once by one launch and read once by the next
//                      (what the per-layer kernels do today).
//
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/l2_handoff.hip -o scripts/probes/l2_handoff
// run:   l2_handoff [NT=2048] [R=8] [PLAIN=0]   (prints GB/s of both modes; wrap in rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
//        PLAIN=1: plain payload stores + agent release fence, agent acquire + plain loads (lines stay in the XCD's L2)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr int TILE = 16384, THREADS = 512;      // 32 B per thread per tile
constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ void store_sc1(u32x4* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 load_sc1(const u32x4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned flag_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// waits until *p >= want (one lane polls, the result is broadcast through LDS); false on timeout or abort
__device__ bool wait_flag(const unsigned* p, unsigned want, int* abort_flag, unsigned* sh) {
  if (threadIdx.x == 0) {
    unsigned ok = 0;
    for (unsigned it = 0; it < SPIN_MAX; ++it) {
      if (flag_load(p) >= want) { ok = 1; break; }
      if ((it & 1023) == 1023 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *sh = ok;
  }
  __syncthreads();
  const bool ok = *sh != 0;
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(THREADS) void k_handoff(char* ring, unsigned* full, unsigned* freed, unsigned long long* sums,
                                                     int nt, int R, int* abort_flag, int plain) {
  __shared__ unsigned sh;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;      // idx 0..31 inside the XCD group
  const int pair = xcd * (gridDim.x / 16) + (idx >> 1);
  const bool producer = (idx & 1) == 0;
  char* base = ring + (size_t)pair * R * TILE;
  unsigned* fl = full + (size_t)pair * R;
  unsigned* fr = freed + (size_t)pair * R;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int t = 0; t < nt; ++t) {
    const int slot = t % R;
    const unsigned gen = (unsigned)(t / R);
    u32x4* tile = reinterpret_cast<u32x4*>(base + (size_t)slot * TILE) + threadIdx.x * 2;
    if (producer) {
      if (gen > 0 && !wait_flag(&fr[slot], gen, abort_flag, &sh)) return;       // slot handed back gen times
      const u32x4 v = {(unsigned)t, (unsigned)pair, threadIdx.x, 0x9e3779b9u * (unsigned)(t + 1)};
      if (plain) { tile[0] = v; tile[1] = v + 1u; } else { store_sc1(tile, v); store_sc1(tile + 1, v + 1u); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        if (plain) {   // plain stores keep the lines in this XCD's L2: publish with an agent-scope release (buffer_wbl2)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        flag_store(&fl[slot], gen + 1);
      }
    } else {
      if (!wait_flag(&fl[slot], gen + 1, abort_flag, &sh)) return;
      u32x4 a, c;
      if (plain) {     // acquire: invalidate this CU's L1, then plain loads (L2 hits when the producer shares the XCD)
        if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        a = tile[0]; c = tile[1];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else { a = load_sc1(tile); c = load_sc1(tile + 1); }
      acc ^= a ^ c;
      __syncthreads();                                                           // every wave's loads have returned
      if (threadIdx.x == 0) flag_store(&fr[slot], gen + 1);
    }
  }
  if (!producer) {
    unsigned long long s = (unsigned long long)acc.x + acc.y + acc.z + acc.w;
    atomicAdd(&sums[pair], s);
  }
}

__global__ __launch_bounds__(THREADS) void k_write(char* buf, int nt) {
  const int pair = blockIdx.x;
  for (int t = 0; t < nt; ++t) {
    u32x4* tile = reinterpret_cast<u32x4*>(buf + ((size_t)pair * nt + t) * TILE) + threadIdx.x * 2;
    const u32x4 v = {(unsigned)t, (unsigned)pair, threadIdx.x, 0x9e3779b9u * (unsigned)(t + 1)};
    tile[0] = v;
    tile[1] = v + 1u;
  }
}
__global__ __launch_bounds__(THREADS) void k_read(const char* buf, unsigned long long* sums, int nt) {
  const int pair = blockIdx.x;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int t = 0; t < nt; ++t) {
    const u32x4* tile = reinterpret_cast<const u32x4*>(buf + ((size_t)pair * nt + t) * TILE) + threadIdx.x * 2;
    acc ^= tile[0] ^ tile[1];
  }
  unsigned long long s = (unsigned long long)acc.x + acc.y + acc.z + acc.w;
  atomicAdd(&sums[pair], s);
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 2048, R = argc > 2 ? atoi(argv[2]) : 8, plain = argc > 3 ? atoi(argv[3]) : 0;
  const int grid = 256, pairs = grid / 2;
  char *ring, *buf;
  unsigned *full, *freed;
  unsigned long long *s0, *s1;
  int* abort_flag;
  const size_t total = (size_t)pairs * nt * TILE;
  CHK(hipMalloc(&ring, (size_t)pairs * R * TILE));
  CHK(hipMalloc(&buf, total));
  CHK(hipMalloc(&full, pairs * R * 4)); CHK(hipMalloc(&freed, pairs * R * 4));
  CHK(hipMalloc(&s0, pairs * 8)); CHK(hipMalloc(&s1, pairs * 8)); CHK(hipMalloc(&abort_flag, 4));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    CHK(hipMemset(full, 0, pairs * R * 4)); CHK(hipMemset(freed, 0, pairs * R * 4));
    CHK(hipMemset(s0, 0, pairs * 8)); CHK(hipMemset(s1, 0, pairs * 8)); CHK(hipMemset(abort_flag, 0, 4));
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_handoff, dim3(grid), dim3(THREADS), 0, 0, ring, full, freed, s0, nt, R, abort_flag, plain);
    hipEventRecord(e1);
    CHK(hipDeviceSynchronize());
    float ms_h = 0; hipEventElapsedTime(&ms_h, e0, e1);
    int ab = 0; CHK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_write, dim3(pairs), dim3(THREADS), 0, 0, buf, nt);
    hipLaunchKernelGGL(k_read, dim3(pairs), dim3(THREADS), 0, 0, buf, s1, nt);
    hipEventRecord(e1);
    CHK(hipDeviceSynchronize());
    float ms_b = 0; hipEventElapsedTime(&ms_b, e0, e1);
    unsigned long long h0[128], h1[128];
    CHK(hipMemcpy(h0, s0, pairs * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h1, s1, pairs * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < pairs; ++i) bad += h0[i] != h1[i];
    printf("rep %d: %d pairs x %d tiles x 16 KiB = %.2f GB handed over (ring %d slots = %.1f MB in all)\n", rep, pairs, nt,
           total / 1e9, R, pairs * R * TILE / 1e6);
    printf("   hand-off through the ring (%s): %.3f ms = %.0f GB/s produced+consumed, timeout/abort %d, checksum mismatches %d\n",
           plain ? "plain stores + release / acquire + plain loads" : "sc1 stores / sc1 loads", ms_h, total / (ms_h * 1e-3) / 1e9, ab, bad);
    printf("   write launch + read launch: %.3f ms = %.0f GB/s (128 workgroups each)\n", ms_b, total / (ms_b * 1e-3) / 1e9);
  }
  return 0;
}
