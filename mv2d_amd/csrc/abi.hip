// C-ABI plumbing shared by all entry points: thread-local last-error string, version, device query.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void mv2d_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* mv2d_last_error(void) { return g_err; }

extern "C" int mv2d_abi_version(void) { return 6; }      // 6 (round 6): e4m3 lo rows (lo_fmt arguments of mv2d_pe_fused_x3 / x3b, mv2d_xattn_tile_fwd_ordered, mv2d_xattn_fused_fwd; mv2d_roi_align_ex +out*_lo8); 5 (rounds 5-6): mv2d_roi_positions_csr (+order_flags), packed x3 weights = fp16 pairs, mv2d_xattn_group_*; 4 (round 4): key16 = fp16 operands, mv2d_decode_topk (+payload) / mv2d_xattn_query_order (+stride) changed, retired entries removed; 3: tile cross attention (mv2d_xattn_*); 2: batches of samples (grp_start arguments), bf16x3 linears / heads, attention backward

// returns the gfx arch name of the current device into buf (e.g. "gfx950:sramecc+:xnack-"); 0 on success
extern "C" int mv2d_device_arch(char* buf, int buflen) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        mv2d_set_error("mv2d_device_arch: no HIP device");
        return MV2D_ERR_LAUNCH;
    }
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return MV2D_OK;
}

// Calibration kernel for the stream planner (mv2d_amd/streams.py): one wave that spins for `usec` microseconds of the
// constant 100 MHz wall clock.  Two HIP streams whose hardware queues really run side by side finish N of these in the time
// of N, streams multiplexed onto one queue/pipe take 2N.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int mv2d_spin(int usec, void* stream) {
    MV2D_CHECK_ARG(usec >= 0 && usec <= 100000, "mv2d_spin: usec must be in [0, 100000]");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)usec * 100);
    MV2D_LAUNCH_CHECK();
    return MV2D_OK;
}
