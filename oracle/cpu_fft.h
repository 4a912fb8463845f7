/* TEST INFRASTRUCTURE (oracle). Deterministic single-precision forward complex FFT
 * used (a) by the fftw3f stand-in that lets the unmodified reference link
 * (oracle/ref_shim/fftw3.h) and (b) by the C restatement oracle/nrsc5_oracle.c.
 * The reference calls FFTW 3.3.x (acquire.c:315-320,254); FFTW is absent from
 * /root/reference and from this image, and the reference's tests never pin FFT
 * output bits, so this stage is tolerance-checked ("parity unpinned" at the FFT
 * boundary). */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* out-of-place, interleaved re/im, n in {256, 2048} (any 4^a*2^b <= 4096), e^{-2 pi i jk/n} */
void oracle_fft_forward(int n, const float *in, float *out);
#ifdef __cplusplus
}
#endif
