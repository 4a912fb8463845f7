// K1 -- front end for gfx950: cu8 IQ -> Q15 -> 15-tap half-band, 2:1 (FM).
// Replaces decimate_samples (input.c:52-69) + halfband_q15_execute / dotprod_halfband_4
// (firdecim_q15.c:137-165, generic branch): every product is shifted right by 15 before it is
// added into an int16 accumulator that wraps, exactly as the reference's integer code does.
//
// Memory-bound stage: each lane pulls 16 B (8 complex u8 samples) per vector load and emits 16 B
// (4 complex Q15 samples); the 14-sample look-back comes from the two preceding 16-B words, which
// neighbouring lanes have just brought into L1, so HBM sees every input byte once.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace nrsc5 {

__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }

__device__ inline int q15_of_u8(unsigned x) { return ((int)x - 127) * 64; }             // U8_Q15, defines.h:93

// one output component from the 15-sample window a[0..14]
__device__ inline int hb_dot(const int *a, int t0, int t1, int t2, int t3)
{
    int acc = 0;
    acc = (int16_t)(acc + (((a[0] + a[14]) * t0) >> 15));
    acc = (int16_t)(acc + (((a[2] + a[12]) * t1) >> 15));
    acc = (int16_t)(acc + (((a[4] + a[10]) * t2) >> 15));
    acc = (int16_t)(acc + (((a[6] + a[8]) * t3) >> 15));
    return (int16_t)(acc + a[7]);
}

// sample k of the virtual stream  [14 history samples | chunk]: k in [-14, nsamples)
__device__ inline void fetch_q15(const uint8_t *iq, const c16 *hist, long long k, int &r, int &i)
{
    if (k < 0) { r = hist[14 + k].r; i = hist[14 + k].i; }
    else { r = q15_of_u8(iq[2 * k]); i = q15_of_u8(iq[2 * k + 1]); }
}

// End of a chunk of nsamp complex input samples (one work-item per stream): note what decim[0]'s last compaction inside the chunk
// leaves at the front of its window (StaleWindows, nrsc5_dev.h), then roll the 14-sample history.
__device__ inline void hb_roll_history(StreamState &st, const uint8_t *iq, long long nsamp)
{
    const long long p = stale_start(st.stale.hb_pushed, nsamp, 14);
    c16 nh[14], sw[14];
    for (int k = 0; k < 14; k++) {
        int r, i;
        fetch_q15(iq, st.hb_hist, nsamp - 14 + k, r, i);
        nh[k].r = (int16_t)r; nh[k].i = (int16_t)i;
        if (p != STALE_NONE) { fetch_q15(iq, st.hb_hist, p + k, r, i); sw[k].r = (int16_t)r; sw[k].i = (int16_t)i; }
    }
    for (int k = 0; k < 14; k++) {
        st.hb_hist[k] = nh[k];
        if (p != STALE_NONE) st.stale.hb[k] = sw[k];
    }
    st.stale.hb_pushed += nsamp;
}

__global__ __launch_bounds__(256) void k_decimate_fm_cu8(DevTables tb, DevBuffers db, const int *ids,
                                                         const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes)
{
    const int sidx = blockIdx.y;
    const int s = stream_of(ids, sidx);
    const StreamState &st = db.state[s];
    const unsigned nout = nbytes[sidx] / 4;                    // outputs in this chunk
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 outputs
    if (4u * g >= nout) return;
    const uint8_t *iq = iq_base + (size_t)sidx * iq_stride;
    c16 *out = db.q15 + (size_t)s * db.q15_cap + (st.wr - st.base);
    const int t0 = tb.hb_q15[0], t1 = tb.hb_q15[1], t2 = tb.hb_q15[2], t3 = tb.hb_q15[3];

    // outputs m = 4g..4g+3 need samples 8g-14 .. 8g+6 (output m: window [2m-14, 2m])
    int wr_[21], wi_[21];
    const bool fast = (g >= 2) && (4u * g + 4u <= nout) && ((((size_t)iq) & 15) == 0);
    if (fast) {
        const uint4 *v = (const uint4 *)iq + g;                // 16 B = samples 8g .. 8g+7
        const uint4 a = v[-2], b = v[-1], c = v[0];            // samples 8g-16 .. 8g+7
        const unsigned w[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
#pragma unroll
        for (int k = 0; k < 21; k++) {                         // sample 8g-14+k = word-stream sample k+2
            const int q = k + 2;
            const unsigned pair = (w[q >> 1] >> ((q & 1) * 16)) & 0xffffu;
            wr_[k] = q15_of_u8(pair & 0xff); wi_[k] = q15_of_u8(pair >> 8);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 21; k++) {
            long long idx = (long long)8 * g - 14 + k;
            if (idx < (long long)2 * nout) fetch_q15(iq, st.hb_hist, idx, wr_[k], wi_[k]);
            else { wr_[k] = 0; wi_[k] = 0; }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        if (4u * g + m < nout) {
            c16 y;
            y.r = (int16_t)hb_dot(wr_ + 2 * m, t0, t1, t2, t3);
            y.i = (int16_t)hb_dot(wi_ + 2 * m, t0, t1, t2, t3);
            out[4 * g + m] = y;
        }
    }
}

// After the chunk: roll the 14-sample history and publish the new write pointer.
__global__ void k_decimate_commit(DevBuffers db, const int *ids, const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes, int nstreams)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nstreams) return;
    const int s = stream_of(ids, sidx);
    StreamState &st = db.state[s];
    const uint8_t *iq = iq_base + (size_t)sidx * iq_stride;
    const long long nsamp = nbytes[sidx] / 2;                  // complex input samples (even)
    if (nsamp == 0) return;
    hb_roll_history(st, iq, nsamp);
    st.wr += nsamp / 2;
}

// Streaming seam, one stream: the chunk is read WHERE THE HOST STAGED IT -- pinned, device-visible memory; a load from it crosses
// PCIe, so every 16-byte word is fetched exactly once (by the lane that owns it, plus a two-word halo per workgroup) and handed
// to its two right-hand neighbours through LDS.  No copy engine in the chain (hipMemcpyAsync + the dependency between the SDMA queue
// and the compute queue cost ~10 us + 2 x ~9 us per block, profiles/r04_dropin_timeline.txt), no second device buffer.  The
// workgroup that finishes LAST rolls the 14-sample history and publishes the new write position (k_decimate_commit's job: every
// other workgroup has read st.wr / st.hb_hist by then), so the chunk costs one launch.
// (the byte count is a kernel argument: read from the pinned header it was one more PCIe round trip in front of every load)
__global__ __launch_bounds__(256) void k_decimate_fm_cu8_stream(DevTables tb, DevBuffers db, int s, const uint8_t *iq, unsigned nb, unsigned *ticket)
{
    StreamState &st = db.state[s];
    const unsigned nout = nb / 4;                              // outputs in this chunk
    const unsigned g0 = blockIdx.x * 256u, g = g0 + threadIdx.x;   // group of 4 outputs = input word g (16 bytes: samples 8g .. 8g+7)
    __shared__ uint4 tile[256 + 2];
    __shared__ int last_block;
    const uint4 *v = (const uint4 *)iq;
    const bool whole = 16ull * (g + 1) <= nb;                  // the word lies inside the chunk
    if (whole) tile[threadIdx.x + 2] = v[g];
    if (threadIdx.x < 2 && g0 >= 2) tile[threadIdx.x] = v[g0 - 2 + threadIdx.x];
    __syncthreads();
    {
        // decim[0]'s last compaction inside this chunk (StaleWindows, nrsc5_dev.h): the 14 samples in front of it are taken where they already are -- in the tile of
        // the workgroup that owns the last of them (its two-word halo reaches the first) -- rather than fetched across PCIe once more by the chunk's last workgroup
        const long long p = stale_start(st.stale.hb_pushed, nb / 2, 14);
        const long long q = p + 13;                            // the last of the 14, chunk-relative (>= -1)
        const unsigned owner = (p == STALE_NONE || q < 0) ? 0u : (unsigned)(q >> 3) >> 8;
        if (p != STALE_NONE && blockIdx.x == owner && threadIdx.x < 14) {
            const long long k = p + threadIdx.x;
            int r, i;
            if (k < 0 || 16ull * ((k >> 3) + 1) > nb) fetch_q15(iq, st.hb_hist, k, r, i);      // history, or a ragged last word the tile does not hold
            else {
                const uint32_t *w = (const uint32_t *)&tile[(k >> 3) - g0 + 2];
                const unsigned pair = (w[(k & 7) >> 1] >> ((k & 1) * 16)) & 0xffffu;
                r = q15_of_u8(pair & 0xff); i = q15_of_u8(pair >> 8);
            }
            st.stale.hb[threadIdx.x].r = (int16_t)r; st.stale.hb[threadIdx.x].i = (int16_t)i;
        }
        // ... and the chunk's own last 14 samples (the next chunk's history) likewise: parked in hb_next by the workgroup whose tile holds them, moved to hb_hist by
        // the chunk's last workgroup (hb_hist itself is still being read by workgroup 0)
        const long long nsamp = nb / 2, ql = nsamp - 1;
        if (nsamp > 0 && blockIdx.x == ((unsigned)(ql >> 3) >> 8) && threadIdx.x >= 32 && threadIdx.x < 46) {
            const long long k = nsamp - 14 + (threadIdx.x - 32);
            int r, i;
            if (k < 0 || 16ull * ((k >> 3) + 1) > nb) fetch_q15(iq, st.hb_hist, k, r, i);
            else {
                const uint32_t *w = (const uint32_t *)&tile[(k >> 3) - g0 + 2];
                const unsigned pair = (w[(k & 7) >> 1] >> ((k & 1) * 16)) & 0xffffu;
                r = q15_of_u8(pair & 0xff); i = q15_of_u8(pair >> 8);
            }
            st.hb_next[threadIdx.x - 32].r = (int16_t)r; st.hb_next[threadIdx.x - 32].i = (int16_t)i;
        }
    }
    if (4u * g < nout) {
        c16 *out = db.q15 + (size_t)s * db.q15_cap + (st.wr - st.base);
        const int t0 = tb.hb_q15[0], t1 = tb.hb_q15[1], t2 = tb.hb_q15[2], t3 = tb.hb_q15[3];
        int wr_[21], wi_[21];
        if (g >= 2 && whole) {
            const uint4 a = tile[threadIdx.x], b = tile[threadIdx.x + 1], c = tile[threadIdx.x + 2];   // samples 8g-16 .. 8g+7
            const unsigned w[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
#pragma unroll
            for (int k = 0; k < 21; k++) {                     // sample 8g-14+k = word-stream sample k+2
                const int q = k + 2;
                const unsigned pair = (w[q >> 1] >> ((q & 1) * 16)) & 0xffffu;
                wr_[k] = q15_of_u8(pair & 0xff); wi_[k] = q15_of_u8(pair >> 8);
            }
        } else {                                               // the chunk's first two groups (history) and a ragged last one
#pragma unroll
            for (int k = 0; k < 21; k++) {
                long long idx = (long long)8 * g - 14 + k;
                if (idx < (long long)2 * nout) fetch_q15(iq, st.hb_hist, idx, wr_[k], wi_[k]);
                else { wr_[k] = 0; wi_[k] = 0; }
            }
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (4u * g + m < nout) {
                c16 y;
                y.r = (int16_t)hb_dot(wr_ + 2 * m, t0, t1, t2, t3);
                y.i = (int16_t)hb_dot(wi_ + 2 * m, t0, t1, t2, t3);
                out[4 * g + m] = y;
            }
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_block = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!last_block || threadIdx.x != 0) return;
    __threadfence();
    *ticket = 0;
    const long long nsamp = nb / 2;                            // complex input samples (even)
    if (nsamp == 0) return;
    for (int k = 0; k < 14; k++) st.hb_hist[k] = st.hb_next[k];   // (parked above, as the compaction was noted above: nothing crosses PCIe here)
    st.stale.hb_pushed += nsamp;
    st.wr += nsamp / 2;
}

void launch_decimate_fm_cu8_stream(const DevTables &tb, const DevBuffers &db, int s, const uint8_t *iq, unsigned nbytes, unsigned *ticket, hipStream_t st)
{
    const unsigned groups = (nbytes / 4 + 3) / 4;
    hipLaunchKernelGGL(k_decimate_fm_cu8_stream, dim3(groups ? (groups + 255) / 256 : 1), dim3(256), 0, st, tb, db, s, iq, nbytes, ticket);
}

void launch_decimate_fm_cu8(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids,
                            const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes, unsigned max_nbytes,
                            hipStream_t st)
{
    const unsigned groups = (max_nbytes / 4 + 3) / 4;
    if (groups) {
        dim3 grid((groups + 255) / 256, nstreams);
        hipLaunchKernelGGL(k_decimate_fm_cu8, grid, dim3(256), 0, st, tb, db, stream_ids, iq_base, iq_stride, nbytes);
    }
    hipLaunchKernelGGL(k_decimate_commit, dim3((nstreams + 63) / 64), dim3(64), 0, st, db, stream_ids, iq_base, iq_stride, nbytes, nstreams);
}

// engine option batch_zero_copy: a freshly reset stream takes the caller's capture as is -- nothing is copied; the symbol
// kernel (k_mixfft) and the acquisition (k_acq_decimate) run the half-band on the fly (halfband_raw.h)
__global__ void k_attach_raw(DevBuffers db, const int *ids, const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes, int nstreams)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nstreams) return;
    StreamState &st = db.state[stream_of(ids, sidx)];
    st.raw = iq_base + (size_t)sidx * iq_stride;
    st.wr = nbytes[sidx] / 4;                                  // decimated samples the capture holds
    st.base = 0;
}

void launch_attach_raw(const DevBuffers &db, int nstreams, const int *stream_ids, const uint8_t *iq_base, long long iq_stride,
                       const unsigned *nbytes, hipStream_t st)
{
    hipLaunchKernelGGL(k_attach_raw, dim3((nstreams + 63) / 64), dim3(64), 0, st, db, stream_ids, iq_base, iq_stride, nbytes, nstreams);
}

// cs16 input is already at 744187.5 S/s: it bypasses the decimator (input.c:119-124)
__global__ __launch_bounds__(256) void k_append_cs16(DevBuffers db, const int *ids, const int16_t *iq_base, long long iq_stride, const unsigned *nsamples)
{
    const int sidx = blockIdx.y;
    const int s = stream_of(ids, sidx);
    const StreamState &st = db.state[s];
    const unsigned n = nsamples[sidx] / 2;
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const c16 *in = (const c16 *)(iq_base + (size_t)sidx * iq_stride);
    db.q15[(size_t)s * db.q15_cap + (st.wr - st.base) + k] = in[k];
}
__global__ void k_append_commit(DevBuffers db, const int *ids, const unsigned *nsamples, int nstreams)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nstreams) return;
    db.state[stream_of(ids, sidx)].wr += nsamples[sidx] / 2;
}

void launch_append_cs16(const DevBuffers &db, int nstreams, const int *stream_ids,
                        const int16_t *iq_base, long long iq_stride, const unsigned *nsamples, unsigned max_n, hipStream_t st)
{
    if (max_n / 2) {
        dim3 grid((max_n / 2 + 255) / 256, nstreams);
        hipLaunchKernelGGL(k_append_cs16, grid, dim3(256), 0, st, db, stream_ids, iq_base, iq_stride, nsamples);
    }
    hipLaunchKernelGGL(k_append_commit, dim3((nstreams + 63) / 64), dim3(64), 0, st, db, stream_ids, nsamples, nstreams);
}

}  // namespace nrsc5
