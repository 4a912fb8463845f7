// K2 -- coarse acquisition for gfx950 (streams whose sync state is not FINE):
//   a) 32-tap Q15 band-select FIR over the 33-symbol window      (firdecim_q15.c:95-109, acquire.c:122-127)
//   b) cyclic-prefix autocorrelation, lag 2048, summed over 32 symbols        (acquire.c:129-134)
//   c) pulse-weighted 112-tap sliding sum, arg-max -> samperr; peak -> angle  (acquire.c:136-151)
// and the per-block bookkeeping that precedes the FFTs (acquire.c:110-119,153-168): k_prepare.
//
// Float accumulation keeps the reference's order (sequential over symbols, then over the 112 taps)
// and the build uses -ffp-contract=off, so the arg-max sees the same values as the CPU path up to
// the FIR (exact) and libm (not used here).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "wave_ops.h"
#include "prepare_block.h"
#include "halfband_raw.h"

namespace nrsc5 {

__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }

__device__ inline bool needs_coarse(const StreamState &st) { return window_ready(st) && st.sync_state != SYNC_FINE; }

// acquisition window of a stream: the FIFO at rd, or -- zero-copy batch -- the window decimated by k_acq_decimate
__device__ inline const c16 *acq_window(const DevBuffers &db, const StreamState &st, int s)
{
    return st.raw ? db.acq_win + (size_t)s * WIN_N : db.q15 + (size_t)s * db.q15_cap + (st.rd - st.base);
}

// ---- zero-copy batch: decimate the 33-symbol window of an un-synchronised stream out of its cu8 capture ----------
// ---- which streams of the set need the acquisition kernels this step ----------------------------------------------------
// Most steps that run them at all do so for a handful of streams (the ones replaying after a lost lock), so the wide kernels
// walk a compacted list with a fixed small grid instead of dispatching 279 empty workgroups per listed stream.
constexpr int ACQ_ROWS = 32;                                    // grid.y of the two window-wide kernels; row r serves list entries r, r + 32, ...

__global__ __launch_bounds__(256) void k_acq_list(DevBuffers db, const int *ids, int nstreams)
{
    __shared__ int count;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < nstreams; k += 256) {
        const int s = stream_of(ids, k);
        if (needs_coarse(db.state[s])) db.acq_list[atomicAdd(&count, 1)] = s;      // order is irrelevant: streams are independent
    }
    __syncthreads();
    if (threadIdx.x == 0) db.acq_list[db.nstreams_alloc] = count;
}

// ---- zero-copy batch: decimate the 33-symbol window of an un-synchronised stream out of its cu8 capture ----------
__global__ __launch_bounds__(256) void k_acq_decimate(DevTables tb, DevBuffers db)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= WIN_N) return;
    const int n = db.acq_list[db.nstreams_alloc];
    const HbTaps taps = hb_taps(tb.hb_q15);
    for (int k = blockIdx.y; k < n; k += ACQ_ROWS) {
        const int s = db.acq_list[k];
        const StreamState &st = db.state[s];
        if (!st.raw) continue;
        db.acq_win[(size_t)s * WIN_N + t] = hb_sample_q15(st.raw, st.rd + t, taps);
    }
}

// ---- a) FIR ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_acq_fir(DevTables tb, DevBuffers db)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= WIN_N) return;
    const int n = db.acq_list[db.nstreams_alloc];
    for (int k = blockIdx.y; k < n; k += ACQ_ROWS) {
        const int s = db.acq_list[k];
        const StreamState &st = db.state[s];
        const c16 *win = acq_window(db, st, s);
        // a[k] = sample t-31+k; indices < 0 come from the filter's carried history
        int sr = 0, si = 0;
#pragma unroll
        for (int i = 1; i < 16; i++) {
            const int ka = t - 31 + i, kb = t - 31 + (32 - i);
            const c16 xa = ka >= 0 ? win[ka] : st.fir_hist[31 + ka];
            const c16 xb = kb >= 0 ? win[kb] : st.fir_hist[31 + kb];
            const int q = tb.acq_q15[i];
            sr = (int16_t)(sr + (((xa.r + xb.r) * q) >> 15));
            si = (int16_t)(si + (((xa.i + xb.i) * q) >> 15));
        }
        {
            const int kc = t - 15;
            const c16 xc = kc >= 0 ? win[kc] : st.fir_hist[31 + kc];
            const int q = tb.acq_q15[16];
            sr = (int16_t)(sr + ((xc.r * q) >> 15));
            si = (int16_t)(si + ((xc.i * q) >> 15));
        }
        c16 y; y.r = (int16_t)sr; y.i = (int16_t)si;
        db.acq_filt[(size_t)s * WIN_N + t] = y;
    }
}

__device__ inline float2 q15_conj_f(c16 v) { return make_float2((float)v.r / 32767.0f, (float)v.i / -32767.0f); }   // defines.h:111

// ---- b) CP correlation -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_acq_corr(DevBuffers db, const int *ids)
{
    const int s = stream_of(ids, blockIdx.y);
    const StreamState &st = db.state[s];
    if (!needs_coarse(st)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SYM_N) return;
    const c16 *f = db.acq_filt + (size_t)s * WIN_N;
    float sr = 0.0f, si = 0.0f;
    for (int j = 0; j < NSYM; j++) {
        const float2 a = q15_conj_f(f[i + j * SYM_N]);
        float2 b = q15_conj_f(f[i + j * SYM_N + FFT_N]);
        b.y = -b.y;                                            // conjf
        const float pr = a.x * b.x - a.y * b.y;
        const float pi = a.x * b.y + a.y * b.x;
        sr += pr; si += pi;
    }
    db.acq_sums[(size_t)s * SYM_N + i] = make_float2(sr, si);
}

// ---- c) weighted sliding sum + arg-max ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_acq_peak(DevTables tb, DevBuffers db, const int *ids)
{
    const int s = stream_of(ids, blockIdx.x);
    StreamState &st = db.state[s];
    if (!needs_coarse(st)) return;                             // block-uniform
    __shared__ float2 sums[SYM_N];
    __shared__ float sh_a[CP_N], sh_b[CP_N];
    __shared__ float red_mag[4]; __shared__ int red_idx[4]; __shared__ float2 red_v[4];
    const int tid = threadIdx.x;
    for (int i = tid; i < SYM_N; i += 256) sums[i] = db.acq_sums[(size_t)s * SYM_N + i];
    for (int j = tid; j < CP_N; j += 256) { sh_a[j] = tb.shape[j]; sh_b[j] = tb.shape[j + FFT_N]; }
    __syncthreads();

    float best_mag = -1.0f; int best_i = 0x7fffffff; float2 best_v = make_float2(0.0f, 0.0f);
    for (int i = tid; i < SYM_N; i += 256) {
        float vr = 0.0f, vi = 0.0f;
        int k = i;
        for (int j = 0; j < CP_N; j++) {
            const float2 z = sums[k];
            vr += (z.x * sh_a[j]) * sh_b[j];
            vi += (z.y * sh_a[j]) * sh_b[j];
            if (++k == SYM_N) k = 0;
        }
        const float mag = vr * vr + vi * vi;
        if (mag > best_mag) { best_mag = mag; best_i = i; best_v = make_float2(vr, vi); }   // ascending i per lane
    }
    // first maximum in index order wins (strict > in the reference's scan)
    for (int m = 32; m >= 1; m >>= 1) {
        const float om = __shfl_xor(best_mag, m); const int oi = __shfl_xor(best_i, m);
        const float ox = __shfl_xor(best_v.x, m), oy = __shfl_xor(best_v.y, m);
        if (om > best_mag || (om == best_mag && oi < best_i)) { best_mag = om; best_i = oi; best_v = make_float2(ox, oy); }
    }
    if ((tid & 63) == 0) { red_mag[tid >> 6] = best_mag; red_idx[tid >> 6] = best_i; red_v[tid >> 6] = best_v; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++)
            if (red_mag[w] > best_mag || (red_mag[w] == best_mag && red_idx[w] < best_i)) { best_mag = red_mag[w]; best_i = red_idx[w]; best_v = red_v[w]; }
        st.coarse_samperr = (best_i + SYM_N - 15) % SYM_N;     // FILTER_DELAY, acquire.c:149
        st.coarse_re = best_v.x; st.coarse_im = best_v.y;
    }
    // the FIR's sliding window now ends at the last sample of this acquire window
    if (tid < 31) {
        const c16 *win = acq_window(db, st, s);
        st.fir_hist[tid] = win[WIN_N - 31 + tid];
        const long long p = stale_start(st.stale.fir_pushed[MODE_FM], WIN_N, 31);          // filter_fm's last compaction inside this block (StaleWindows, nrsc5_dev.h)
        if (p != STALE_NONE) st.stale.fir[MODE_FM][tid] = win[p + tid];
    }
    __syncthreads();
    if (tid == 0) st.stale.fir_pushed[MODE_FM] += WIN_N;
}

void launch_acquire(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st)
{
    const int rows = nstreams < ACQ_ROWS ? nstreams : ACQ_ROWS;
    hipLaunchKernelGGL(k_acq_list, dim3(1), dim3(256), 0, st, db, stream_ids, nstreams);
    if (db.acq_win) hipLaunchKernelGGL(k_acq_decimate, dim3((WIN_N + 255) / 256, rows), dim3(256), 0, st, tb, db);
    hipLaunchKernelGGL(k_acq_fir, dim3((WIN_N + 255) / 256, rows), dim3(256), 0, st, tb, db);
    hipLaunchKernelGGL(k_acq_corr, dim3((SYM_N + 255) / 256, nstreams), dim3(256), 0, st, db, stream_ids);
    hipLaunchKernelGGL(k_acq_peak, dim3(nstreams), dim3(256), 0, st, tb, db, stream_ids);
}

// ---- per-block bookkeeping: see prepare_block.h ---------------------------------------------------------
__global__ void k_prepare(DevBuffers db, const int *ids, int nstreams, int acq_on)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nstreams) return;
    const int s = stream_of(ids, sidx);
    prepare_block(db, db.state[s], s, acq_on != 0);
}

void launch_prepare(const DevBuffers &db, int nstreams, const int *stream_ids, int acq_on, hipStream_t st)
{
    hipLaunchKernelGGL(k_prepare, dim3((nstreams + 63) / 64), dim3(64), 0, st, db, stream_ids, nstreams, acq_on);
}

}  // namespace nrsc5
