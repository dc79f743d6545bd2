// Device-wide barrier cost on MI355X (8 XCDs, one L2 each): how should a persistent kernel synchronise its
// workgroups between two phases?  Variants, each run ITERS times by a cooperative launch of G workgroups x 256:
//   flat   every workgroup does one agent-scope atomic add on ONE counter and spins on it
//   tree   16 workgroups share a first-level counter, the last arriver of each group bumps the root
//   flags  no atomics: every workgroup stores its own arrival flag; workgroup 0 polls all flags with 256 lanes and
//          then stores one release flag PER workgroup; every workgroup polls only its own release flag
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gridsync.hip -o /tmp/ubench_gridsync
// Result (MI355X, 256 CUs), microseconds per barrier: see DESIGN.md section 5 (LSM).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ unsigned ld(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_flat(unsigned* bar, int iters, unsigned* fail) {
    unsigned epoch = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            epoch += gridDim.x;
            __threadfence();
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (ld(bar) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) {
                    st(fail, 1u);
                    break;
                }
            }
            __threadfence();
        }
        __syncthreads();
    }
}

// bar[0] = root, bar[16 + g] = first-level counter of group g (16 workgroups each)
__global__ __launch_bounds__(256) void k_tree(unsigned* bar, int iters, unsigned* fail) {
    const unsigned G = gridDim.x, grp = blockIdx.x >> 4, n_grp = (G + 15) >> 4;
    const unsigned grp_size = (grp == n_grp - 1) ? G - (grp << 4) : 16u;
    unsigned epoch = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            ++epoch;
            __threadfence();
            const unsigned t = __hip_atomic_fetch_add(bar + 16 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == epoch * grp_size) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (ld(bar) < epoch * n_grp) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) {
                    st(fail, 1u);
                    break;
                }
            }
            __threadfence();
        }
        __syncthreads();
    }
}

// arrive[b], go[b]
__global__ __launch_bounds__(256) void k_flags(unsigned* arrive, unsigned* go, int iters, unsigned* fail) {
    const unsigned G = gridDim.x;
    unsigned epoch = 0;
    for (int it = 0; it < iters; ++it) {
        ++epoch;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            st(arrive + blockIdx.x, epoch);
        }
        if (blockIdx.x == 0) {
            for (unsigned b = threadIdx.x; b < G; b += 256) {
                unsigned spins = 0;
                while (ld(arrive + b) < epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) {
                        st(fail, 1u);
                        break;
                    }
                }
            }
            __syncthreads();
            __threadfence();
            // (the master's reduce + solve would sit here)
            for (unsigned b = threadIdx.x; b < G; b += 256) st(go + b, epoch);
        }
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (ld(go + blockIdx.x) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) {
                    st(fail, 1u);
                    break;
                }
            }
            __threadfence();
        }
        __syncthreads();
    }
}

// flags, without agent-scope fences: everything exchanged goes through sc1 stores/loads (write-through to and read
// from the device coherence point), ordered by s_waitcnt only.  The agent-scope fences above cost a writeback +
// invalidate of the XCD's L2 per workgroup and barrier, and those serialise within the XCD.
__global__ __launch_bounds__(256) void k_flags_nofence(unsigned* arrive, unsigned* go, int iters, unsigned* fail) {
    const unsigned G = gridDim.x;
    unsigned epoch = 0;
    for (int it = 0; it < iters; ++it) {
        ++epoch;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_s_waitcnt(0);  // earlier sc1 stores of this wave have been acknowledged
            st(arrive + blockIdx.x, epoch);
        }
        if (blockIdx.x == 0) {
            for (unsigned b = threadIdx.x; b < G; b += 256) {
                unsigned spins = 0;
                while (ld(arrive + b) < epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) {
                        st(fail, 1u);
                        break;
                    }
                }
            }
            __syncthreads();
            __builtin_amdgcn_s_waitcnt(0);
            for (unsigned b = threadIdx.x; b < G; b += 256) st(go + b, epoch);
        }
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (ld(go + blockIdx.x) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) {
                    st(fail, 1u);
                    break;
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_flat_nofence(unsigned* bar, int iters, unsigned* fail) {
    unsigned epoch = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            epoch += gridDim.x;
            __builtin_amdgcn_s_waitcnt(0);
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (ld(bar) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) {
                    st(fail, 1u);
                    break;
                }
            }
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    unsigned* buf;
    CHECK(hipMalloc(&buf, 1 << 20));
    unsigned* fail = buf + (1 << 17);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int G : {64, 256, 512, 1024}) {
        for (int variant = 0; variant < 5; ++variant) {
            CHECK(hipMemset(buf, 0, 1 << 20));
            unsigned* a0 = buf;
            unsigned* a1 = buf + 65536;
            int it = iters;
            void* p3[] = {&a0, &it, &fail};
            void* p4[] = {&a0, &a1, &it, &fail};
            const void* fns[5] = {(const void*)k_flat, (const void*)k_tree, (const void*)k_flags, (const void*)k_flags_nofence,
                                  (const void*)k_flat_nofence};
            const char* names[5] = {"flat", "tree", "flags", "flags_nofence", "flat_nofence"};
            const void* fn = fns[variant];
            CHECK(hipEventRecord(e0, 0));
            CHECK(hipLaunchCooperativeKernel(fn, dim3(G), dim3(256), (variant == 2 || variant == 3) ? p4 : p3, 0, 0));
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned f = 0;
            CHECK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            std::printf("G=%4d %-14s %7.2f us/barrier%s\n", G, names[variant],
                        ms * 1e3 / iters, f ? "  (TIMED OUT)" : "");
        }
    }
    return 0;
}
