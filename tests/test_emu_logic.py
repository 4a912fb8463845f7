"""`-m "not gpu"`: the HIP kernel sources compiled with g++ against the SIMT emulator
(tests/simt/) and checked against the oracle.  This validates kernel LOGIC in the GPU-less
container; it is not a product path and proves nothing about gfx950 code generation -- the
`-m gpu` twins in test_gpu_parity.py do that through the real library."""
import os

import pytest

from tests import engine_checks as ec
from nrsc5_amd import synth


def test_emu_halfband(emu_lib, oracle):
    ec.check_halfband(emu_lib, oracle, n=5003)


def test_emu_halfband_history(emu_lib, oracle):
    ec.check_halfband_streaming_history(emu_lib, oracle)


def test_emu_fft(emu_lib, oracle):
    ec.check_fft(emu_lib, oracle, n=2)


def test_emu_viterbi(emu_lib, oracle):
    ec.check_viterbi(emu_lib, oracle, lens=(80, 2304), frames=3)


def test_emu_viterbi_roundtrip(emu_lib):
    ec.check_viterbi_roundtrip(emu_lib, L=1152, frames=2)


def test_emu_end_to_end_golden_cfo_search(emu_lib, captures):
    ec.check_golden_end_to_end(emu_lib, "fm_cu8_cfo-2400", captures)


def test_emu_end_to_end_oracle_short(emu_lib, oracle):
    ec.check_oracle_end_to_end(emu_lib, oracle, dict(n_frames=0, n_blocks=8, seed=21, cfo_hz=-212.0, offset=2501, snr_db=11.0))


def test_emu_batch_equals_streaming(emu_lib):
    caps = [synth.fm_mp1_capture(0, seed=30 + k, cfo_hz=c, offset=o, snr_db=18, n_blocks=6)
            for k, (c, o) in enumerate([(50.0, 100), (-900.0, 3000), (300.0, 0)])]
    ec.check_batch_equals_streaming(emu_lib, caps, p1_async=False)


def test_emu_async_p1_window_pipeline(emu_lib):
    """Frames that complete in the middle of a 16-step decode window (first block spent in COARSE)
    must be gathered exactly once, before the interleaver matrix is refilled."""
    caps = [synth.fm_mp1_capture(0, seed=40 + k, cfo_hz=c, offset=o, snr_db=14, n_blocks=nb)
            for k, (c, o, nb) in enumerate([(0.0, 4319, 20), (350.0, 100, 19)])]
    ec.check_batch_equals_streaming(emu_lib, caps, p1_async=True)


def test_emu_interleaved_streams_and_subset_batches(emu_lib):
    ec.check_interleaved_streams(emu_lib)


def test_emu_zero_copy_batch_equals_streaming(emu_lib):
    """batch_zero_copy: captures read in place, half-band fused into the symbol kernel (incl. a stream that needs the CFO
    search, i.e. the on-the-fly acquisition window)."""
    caps = [synth.fm_mp1_capture(0, seed=70 + k, cfo_hz=c, offset=o, snr_db=18, n_blocks=nb)
            for k, (c, o, nb) in enumerate([(40.0, 123, 20), (-2300.0, 3001, 8)])]
    ec.check_zero_copy_batch(emu_lib, caps)


def test_emu_small_fifo_compaction(emu_lib, captures):
    ec.check_small_fifo_compaction(emu_lib, "fm_cu8_cfo-2400", captures)


def test_emu_api_edges(emu_lib):
    ec.check_api_edges(emu_lib)


def test_emu_large_push_into_minimum_fifo(emu_lib, oracle):
    ec.check_large_push_minimum_fifo(emu_lib, oracle)


def test_emu_extended_sidebands_mp11(emu_lib, oracle):
    """PX1 / PX2 -> interleaver IV -> P3 / P4 (MP11), in-order and through the deferred decode windows."""
    kw = dict(n_frames=0, n_blocks=44, seed=5, mode="MP11", cfo_hz=50.0, offset=500, snr_db=25, fmt="cs16")
    ec.check_oracle_end_to_end(emu_lib, oracle, kw)
    ec.check_oracle_end_to_end(emu_lib, oracle, kw, p1_async=True)


def test_emu_compatibility_modes_5_6(emu_lib, oracle):
    """PSMI 5 / 7 (compatibility modes 5 / 5): 14 partitions equalised and measured, no extended partition routed."""
    for mode in ("PSMI7",):                                    # MP5 / MP6 run on the GPU (test_gpu_compatibility_modes_5_6_oracle)
        log = ec.check_oracle_end_to_end(emu_lib, oracle, dict(n_frames=0, n_blocks=50, seed=41, mode=mode, cfo_hz=35.0, offset=420, snr_db=22))
        assert not [v for k, v in log if k == "frame" and v["lc"] != 0] and any(k == "mer" for k, _ in log)


def test_emu_extended_sidebands_mp2(emu_lib, oracle):
    ec.check_oracle_end_to_end(emu_lib, oracle, dict(n_frames=0, n_blocks=40, seed=6, mode="MP2", cfo_hz=-20.0, offset=777, snr_db=20))


# ---- AM ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lag", [3])
def test_emu_am_replay_equals_reference(emu_lib, oracle, lag):
    """AM window pipeline + on-device L2 feedback == the oracle with the restated frame_process decision (k_rollback_am)."""
    ec.check_am_deferred_feedback_equals_reference(emu_lib, oracle, verdict_lag=lag)


def test_emu_am_reduced_bandwidth(emu_lib, oracle):
    ec.check_am_reduced_bandwidth(emu_lib, oracle)


def test_emu_mixed_batch_pipeline(emu_lib, oracle):
    ec.check_mixed_batch_pipeline(emu_lib, oracle, passes=1)


def test_emu_am_replay_with_cold_segments(emu_lib, oracle):
    """the window pipeline's own decode kernels with warm-up and run-in switched off: every P3 segment boundary takes the repair path"""
    from nrsc5_amd import engine as eng
    ec.check_am_deferred_feedback_equals_reference(emu_lib, oracle, tunes=((eng.TUNE_AM_WARM, 0), (eng.TUNE_AM_SEGMENTS, 5)), expect_k9_repairs=True)


def test_emu_am_viterbi_k9(emu_lib, oracle):
    ec.check_viterbi_k9(emu_lib, oracle, lens=(80, 3750), frames=2)


def test_emu_first_header_check(emu_lib, oracle):
    ec.check_first_header(emu_lib, oracle, seeds=(5,))


def test_emu_am_viterbi_k9_segmented(emu_lib, oracle):
    ec.check_viterbi_k9_segmented(emu_lib, oracle, lens=(3750,), segments=(2, 5))


def test_emu_am_decimator(emu_lib, oracle):
    ec.check_am_decimator(emu_lib, oracle)


def test_emu_am_end_to_end_oracle(emu_lib, oracle):
    ec.check_am_oracle_end_to_end(emu_lib, oracle, dict(n_frames=10, seed=3, cfo_hz=3.0, offset=1234))


def test_emu_am_end_to_end_cu8_ragged(emu_lib, oracle):
    ec.check_am_oracle_end_to_end(emu_lib, oracle, dict(n_frames=2, seed=6, cfo_hz=10.0, offset=64 * 300 + 12, fmt="cu8"), chunk=4100 * 4)


def test_emu_am_batch_equals_streaming(emu_lib):
    ec.check_am_batch_equals_streaming(emu_lib, [dict(n_frames=9, seed=32, cfo_hz=-20.0, offset=5000), dict(n_frames=3, seed=33)])


def test_emu_am_window_pipeline_equals_in_order(emu_lib):
    """p1_async: the nine frames of an L1 frame decode concurrently on a decode stream, BER merged at fetch time."""
    ec.check_am_batch_equals_streaming(emu_lib, [dict(n_frames=11, seed=34, cfo_hz=7.0, offset=300), dict(n_frames=9, seed=35, offset=4000),
                                                 dict(n_frames=9, seed=8, cfo_hz=-6.0, offset=2000, mode="MA3")], p1_async=True)


def test_emu_am_ma3_end_to_end(emu_lib, oracle):
    """All-digital layout (psmi 2): no sideband combining, QAM64 everywhere, 30000-bit P3 frames through the E1 code."""
    ec.check_am_oracle_end_to_end(emu_lib, oracle, dict(n_frames=8, seed=8, cfo_hz=-6.0, offset=2000, mode="MA3"))


def test_emu_l2_feedback_on_device_fm(emu_lib, oracle):
    ec.check_l2_feedback(emu_lib, oracle, dict(n_frames=0, n_blocks=40, seed=23, cfo_hz=0.0, offset=1234, snr_db=20.0))


def test_emu_l2_feedback_on_device_am(emu_lib, oracle):
    ec.check_l2_feedback(emu_lib, oracle, dict(n_frames=16, seed=9, cfo_hz=2.0, offset=500, burst=(8.3, 0.5, 40.0)), am=True)


def test_emu_deferred_feedback_equals_reference(emu_lib, oracle):
    """Window pipeline + on-device L2 feedback (the benchmarked mode): replay makes it reference-identical, here with the
    verdicts taking effect 3 decode windows after their frame (48 speculated blocks are rewound)."""
    ec.check_deferred_feedback_equals_reference(emu_lib, oracle, n_blocks=80, verdict_lag=3, cases=ec.FALSE_LOCK_CASES[:2])


def test_emu_mode_switch_on_live_stream(emu_lib, reflib):
    ec.check_mode_switch(emu_lib, reflib)


def test_emu_reset_of_a_used_stream_keeps_the_fir_windows(emu_lib, reflib):
    """nrsc5hip_stream_reset == the unmodified reference's input_reset on a used session (stale half-band and acquisition-filter windows);
    nrsc5hip_stream_fresh == a new session"""
    ec.check_reset_keeps_fir_windows(emu_lib, reflib)
    ec.check_reset_keeps_fir_windows_am(emu_lib, reflib)


def test_emu_randomised_sessions_of_several_captures(emu_lib, reflib):
    """tools/cpu_session_fuzz.py on fixed seeds: 2 - 4 captures per session (FM cu8 / cs16, AM cs16 / cu8, noise or signal) with nrsc5_set_mode between them; after every
    reset the first decimated samples and the whole log equal the unmodified reference's driven the same way (its FIR windows, sync_t.samperr / .angle / .bc survive
    input_reset).  300 sessions / 885 captures of the tool: no difference that a fresh session of the same capture does not show as well."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cpu_session_fuzz as fz
    for seed in (70001, 70002, 70005, 71008):
        kinds, problems = fz.run_session(emu_lib, reflib, seed)
        assert not [p for p in problems if p[2] == "q15" or "ONLY AFTER" in p[2]], (seed, kinds, problems)


def test_emu_reset_window_compaction_boundaries(emu_lib, reflib):
    ec.check_reset_window_boundaries(emu_lib, reflib)


def test_emu_pids_crc_flag(emu_lib, oracle):
    ec.check_pids_crc_flag(emu_lib, oracle)
    ec.check_pids_crc_flag(emu_lib, oracle, am=True)


def test_emu_l2_index_every_branch(emu_lib, oracle):
    """frame_push + the audio walk of frame_process on the device == the oracle's index (pinned against the reference's
    own frame_push in test_oracle_l2.py), all six frame lengths."""
    ec.check_l2_index_stage(emu_lib, oracle)


def test_emu_l2_index_end_to_end(emu_lib, oracle):
    ec.check_l2_index_end_to_end(emu_lib, oracle, am=True)
    ec.check_l2_index_end_to_end(emu_lib, oracle, am=True, p1_async=True)          # fused index behind the deferred decodes
    ec.check_l2_index_end_to_end(emu_lib, oracle, am=False, mode="MP11", p1_async=True)


def test_emu_l2_index_fused_into_decode(emu_lib, oracle):
    ec.check_l2_index_fused(emu_lib, oracle, p1_async=True)


def test_emu_l2_index_vs_reference_golden(emu_lib):
    ec.check_l2_index_vs_reference_golden(emu_lib)


def test_emu_frame_push_indexed_with_device_index(emu_lib, reflib):
    ec.check_frame_push_indexed_with_device_index(emu_lib, reflib)


def test_emu_block_exact_pushes(emu_lib, oracle):
    ec.check_block_exact_pushes(emu_lib, oracle)


def test_emu_viterbi_segmented(emu_lib, oracle):
    ec.check_viterbi_segmented(emu_lib, oracle, lens=(4608,), segments=(1, 3, 8))


# ---- impaired channels (nrsc5_amd/channel.py): the FINE-state timing feedback at work on every block ----------------------------
@pytest.mark.parametrize("name", ["ppm+60", "ppm-85_cs16", "host_clip_ppm"])
def test_emu_impaired_channel_streaming(emu_lib, oracle, name):
    from tests import common
    ec.check_oracle_end_to_end(emu_lib, oracle, common.IMPAIRED_FM_CASES[name])


def test_emu_impaired_channel_zero_copy_batch(emu_lib):
    """samperr != 0 in FINE moves `keep`, i.e. the position the fused half-band reads the capture from (k_mixfft a0 / st.rd)."""
    from tests import common
    caps = [synth.fm_mp1_capture(**common.IMPAIRED_FM_CASES[n]) for n in ("ppm+100_cfo_search", "fade_ppm", "echoes")]
    ec.check_zero_copy_batch(emu_lib, caps, p1_async=True, l2_feedback=True)


def test_emu_replay_under_drift(emu_lib, oracle):
    """Sync loss 30-50 blocks into a drifting stream, verdict 3 windows late: the checkpoint restores a walked FIFO position."""
    ec.check_deferred_feedback_equals_reference(emu_lib, oracle, n_blocks=96, verdict_lag=3, caps=ec.drift_replay_captures())


def test_emu_am_impaired_channel(emu_lib, oracle):
    from tests import common
    ec.check_am_oracle_end_to_end(emu_lib, oracle, common.IMPAIRED_AM_CASES["am_ppm-50"])


def test_emu_deferred_seam_equals_synchronous(emu_lib):
    ec.check_deferred_seam(emu_lib)


def test_emu_host_capture_seam_equals_fifo_seam(emu_lib):
    ec.check_host_capture_seam(emu_lib, names=("ppm+100_cfo_search",))


@pytest.mark.parametrize("syms", [4, 16])
def test_emu_symbol_kernel_variants(emu_lib, syms):
    """k_mixfft's knob forms -- 4 symbols in a row per workgroup (next-symbol prefetch), 16 = two symbols side by side in a 256-lane
    workgroup -- leave every record as the default form does (zero-copy batch == streaming, rtol 0)"""
    caps = [synth.fm_mp1_capture(0, seed=71, cfo_hz=33.0, offset=400, snr_db=22, n_blocks=20), synth.fm_mp1_capture(0, seed=72, cfo_hz=-120.0, offset=1500, snr_db=20, n_blocks=20)]
    ec.check_zero_copy_batch(emu_lib, caps, p1_async=True, l2_feedback=False, mixfft_syms=syms)


def test_emu_symbol_kernel_256_lanes(emu_lib, oracle, captures):
    """k_mixfft8 (knob 32): the 256-lane symbol kernel -- FFT 2048 = 8 x 8 x 8 x 4 with the last radix across DPP quads -- as an FFT against
    float64, in the zero-copy batch against its own streaming form (rtol 0), and end to end against the reference's golden trace"""
    ec.check_fft(emu_lib, oracle, n=3, form=32)
    caps = [synth.fm_mp1_capture(0, seed=71, cfo_hz=33.0, offset=400, snr_db=22, n_blocks=20), synth.fm_mp1_capture(0, seed=72, cfo_hz=-120.0, offset=1500, snr_db=20, n_blocks=20)]
    ec.check_zero_copy_batch(emu_lib, caps, p1_async=True, l2_feedback=False, mixfft_syms=32, singles_tuned=True)
    ec.check_golden_end_to_end(emu_lib, "fm_cu8_cfo137", captures, tune=((ec.eng.TUNE_MIXFFT_SYMS, 32),))


def test_emu_traceback_variants(emu_lib):
    ec.check_traceback_variants(emu_lib)


def test_emu_exact_oscillator_first_block_is_the_references(emu_lib, reflib):
    ec.check_exact_oscillator_first_block(emu_lib, reflib, bit_exact_min=4, n=4)


def test_emu_dataflow_bursts_equal_the_two_kernel_form(emu_lib):
    """k_flow on the CPU twin (its work items run one after the other in ticket order): three streams -- one through the CFO search, one with a sample-clock error, so
    that bursts of every length and the fall-back to the two-kernel form on acquisition steps are exercised -- record for record, frame for frame."""
    from tests import common
    caps = [synth.fm_mp1_capture(0, seed=81, cfo_hz=33.0, offset=400, snr_db=22, n_blocks=40), synth.fm_mp1_capture(0, seed=82, cfo_hz=-2300.0, offset=3001, snr_db=18, n_blocks=36),
            synth.fm_mp1_capture(**dict(common.IMPAIRED_FM_CASES["ppm+60"], n_blocks=40))]
    ec.check_flow_bursts(emu_lib, caps)
