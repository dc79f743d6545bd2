/* mcgpu_debug.h -- TEST HOOKS of libmcgpu.so.  Not part of the drop-in boundary: a maintainer of the reference binds
 * include/mcgpu.h only.  These entry points let tests/ force conditions a healthy device never shows (a hand-shake that times
 * out, a partial sum that never arrives, a tiny workspace budget) and run the host-only parts (the node segment's join /
 * barrier protocol, the peer-mailbox decision table, single device math routines) on their own.  The symbols are exported by
 * the same library; tests/test_abi_symbols.py checks them like the product's. */
#ifndef MCGPU_DEBUG_H
#define MCGPU_DEBUG_H
#include "mcgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: evaluate one device math routine of the path kernels elementwise (csrc/fastmath.hpp).
 * y has 4n doubles; element i's results start at y[4i].
 * fn 0: y[0] = 1*e^x          fn 1: y[0] = -2 ln x (x in (0,1])       fn 2: y[0] = sqrt(x)
 * fn 3: x holds a 32-bit Philox word wb as a double; y[0], y[1] = cos, sin(2 pi ((wb>>8)+1/2) 2^-24)
 * fn 4: x holds a path id as a double; y[0..3] = the four normals of block 0, stream 0, seed 1
 * fn 5: as 4 through the reference-grade (device library) implementation.
 * fn 6: y[0] = e^x for |x| <= 0.125 (the branch-free step exponential)   fn 7: the same for |x| <= 0.1. */
int mcg_debug_eval(mcg_ctx* ctx, int fn, const double* x, double* y, int64_t n);
/* Test hooks of the one-launch LSM sweeps' hand-shake: spin_limit = polling rounds before a wait gives up (0: every
 * wait gives up at once, which drives the time-out -> per-date fall-back on a healthy device; < 0: the default);
 * poll_delay = units of ~4 us by which workgroups other than the reducing one reach their coefficient poll late. */
int mcg_debug_lsm_hooks(mcg_ctx* ctx, long long spin_limit, int poll_delay);
/* Test hooks of the per-date LSM route's exchange (k_lsm_date): mode 1 = workgroup `workgroup` withholds its partial
 * moments at exercise date `date` (a store that never lands: mcg_price_lsm must fail with MCG_ERR_HIP, never return a
 * price); mode 2 = it sends them `delay` x ~4 us AFTER drawing its ticket (a store that lands late: the consumer waits
 * for it, the price is the usual one); mode 0 = off.  spin_limit = polls before a consumer gives up (< 0: default). */
int mcg_debug_lsm_date_fault(mcg_ctx* ctx, int mode, int date, int workgroup, int delay, long long spin_limit);
/* Test hook: workspace bytes one chunk of mcg_batch_price_rows* may use (0: the default, a quarter of free memory). */
int mcg_debug_batch_budget(mcg_ctx* ctx, size_t bytes);
/* Class-API coalescing (mcg_compat_set_coalescing): at most max_slots calling threads get a matrix slot of the device arena from
 * now on (< 0: the default, 512); a thread that gets none prices on a context of its own.  For the test of that fall-back. */
int mcg_debug_coalesce_slots(int max_slots);
/* The combiner's protocol -- one queue and service thread per kind of call, sleeps and wake-ups, requests queued ahead and taken
 * later or drained, slots, thread exit -- WITHOUT a GPU: n_threads host threads make calls_per_thread calls each, answered by a
 * stand-in for the device that takes ~50 us per round.  *wrong = wrong or missing answers.  Call it in a process that has not
 * used the class API (it switches the combiner to the stand-in for the life of the process). */
int mcg_debug_coalesce_selftest(int n_threads, int calls_per_thread, int* wrong);
/* Test hook, host-only: the decision whether a rank maps a peer's mailbox (1) or all ranks stay on the host mailbox (0),
 * from what it knows about the peer: same process?, does its PCI bus id resolve to a visible device?, the same device?,
 * is peer access available? */
int mcg_debug_peer_decision(int same_process, int bus_id_resolves, int same_device, int can_access_peer);
/* Test hooks, host-only (no GPU needed): the shared segment's protocol on its own -- join `name` as `rank` of
 * `n_ranks` (a stale segment of a crashed job under the same name is never joined), barrier (fails at once on every
 * rank after a time-out or a poison), poison, leave. */
int mcg_debug_shm_attach(const char* name, int n_ranks, int rank, double timeout_s, void** handle);
int mcg_debug_shm_barrier(void* handle);
int mcg_debug_shm_poison(void* handle);
int mcg_debug_shm_detach(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* MCGPU_DEBUG_H */
