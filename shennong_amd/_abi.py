"""ctypes mirror of ``include/shennong_amd.h`` (option structs and constants).

Field meanings and defaults are Kaldi's option structs, which the reference wraps one-to-one
(reference shennong/processor/base.py:122-374, filterbank.py:48-55, mfcc.py:48-56,
plp.py:265-273, pitch_kaldi.py:86-91,321-327, delta.py:54).
"""

import ctypes as C

SNF_OK = 0
SNF_E_INVALID = -1
SNF_E_RUNTIME = -2
SNF_E_HIP = -3
SNF_E_NODEVICE = -4

KIND_SPECTROGRAM = 0
KIND_FBANK = 1
KIND_MFCC = 2
KIND_PLP = 3
KIND_PITCH = 4
KIND_PITCH_POST = 5
KIND_DELTA = 6
KIND_ENERGY = 7
KIND_VAD = 8
KIND_CMVN = 9
KIND_SLIDING_CMVN = 10

WINDOW_TYPES = {
    'hamming': 0, 'hanning': 1, 'povey': 2, 'rectangular': 3, 'blackman': 4}

COMPRESSION = {'off': 0, 'log': 1, 'sqrt': 2}


class FrameOptions(C.Structure):
    _fields_ = [
        ('samp_freq', C.c_float),
        ('frame_shift_ms', C.c_float),
        ('frame_length_ms', C.c_float),
        ('dither', C.c_float),
        ('preemph_coeff', C.c_float),
        ('remove_dc_offset', C.c_int32),
        ('window_type', C.c_int32),
        ('round_to_power_of_two', C.c_int32),
        ('blackman_coeff', C.c_float),
        ('snip_edges', C.c_int32)]


class MelOptions(C.Structure):
    _fields_ = [
        ('num_bins', C.c_int32),
        ('low_freq', C.c_float),
        ('high_freq', C.c_float),
        ('vtln_low', C.c_float),
        ('vtln_high', C.c_float)]


class PitchOptions(C.Structure):
    _fields_ = [
        ('samp_freq', C.c_float),
        ('frame_shift_ms', C.c_float),
        ('frame_length_ms', C.c_float),
        ('preemph_coeff', C.c_float),
        ('min_f0', C.c_float),
        ('max_f0', C.c_float),
        ('soft_min_f0', C.c_float),
        ('penalty_factor', C.c_float),
        ('lowpass_cutoff', C.c_float),
        ('resample_freq', C.c_float),
        ('delta_pitch', C.c_float),
        ('nccf_ballast', C.c_float),
        ('lowpass_filter_width', C.c_int32),
        ('upsample_filter_width', C.c_int32),
        ('recompute_frame', C.c_int32),
        ('snip_edges', C.c_int32)]


class PitchPostOptions(C.Structure):
    _fields_ = [
        ('pitch_scale', C.c_float),
        ('pov_scale', C.c_float),
        ('pov_offset', C.c_float),
        ('delta_pitch_scale', C.c_float),
        ('delta_pitch_noise_stddev', C.c_float),
        ('normalization_left_context', C.c_int32),
        ('normalization_right_context', C.c_int32),
        ('delta_window', C.c_int32),
        ('delay', C.c_int32),
        ('add_pov_feature', C.c_int32),
        ('add_normalized_log_pitch', C.c_int32),
        ('add_delta_pitch', C.c_int32),
        ('add_raw_log_pitch', C.c_int32)]


class VadOptions(C.Structure):
    _fields_ = [
        ('energy_threshold', C.c_float),
        ('energy_mean_scale', C.c_float),
        ('frames_context', C.c_int32),
        ('proportion_threshold', C.c_float)]


class SlidingCmvnOptions(C.Structure):
    _fields_ = [
        ('center', C.c_int32),
        ('cmn_window', C.c_int32),
        ('min_window', C.c_int32),
        ('normalize_variance', C.c_int32)]


class Options(C.Structure):
    _fields_ = [
        ('kind', C.c_int32),
        ('frame', FrameOptions),
        ('mel', MelOptions),
        ('use_energy', C.c_int32),
        ('energy_floor', C.c_float),
        ('raw_energy', C.c_int32),
        ('htk_compat', C.c_int32),
        ('use_log_fbank', C.c_int32),
        ('use_power', C.c_int32),
        ('num_ceps', C.c_int32),
        ('cepstral_lifter', C.c_float),
        ('lpc_order', C.c_int32),
        ('compress_factor', C.c_float),
        ('cepstral_scale', C.c_float),
        ('rasta', C.c_int32),
        ('compression', C.c_int32),
        ('delta_order', C.c_int32),
        ('delta_window', C.c_int32),
        ('append_deltas', C.c_int32),
        ('pitch', PitchOptions),
        ('pitch_post', PitchPostOptions),
        ('vad', VadOptions),
        ('sliding_cmvn', SlidingCmvnOptions),
        ('seed', C.c_uint64)]


def default_frame_options():
    """Kaldi FrameExtractionOptions defaults (reference processor/base.py:122-126)"""
    return FrameOptions(
        samp_freq=16000.0, frame_shift_ms=10.0, frame_length_ms=25.0,
        dither=1.0, preemph_coeff=0.97, remove_dc_offset=1,
        window_type=WINDOW_TYPES['povey'], round_to_power_of_two=1,
        blackman_coeff=0.42, snip_edges=1)


def default_mel_options():
    """Kaldi MelBanksOptions defaults (reference processor/base.py:288-293)"""
    return MelOptions(
        num_bins=23, low_freq=20.0, high_freq=0.0,
        vtln_low=100.0, vtln_high=-500.0)


def default_pitch_options():
    """Kaldi PitchExtractionOptions defaults (reference pitch_kaldi.py:86-91)"""
    return PitchOptions(
        samp_freq=16000.0, frame_shift_ms=10.0, frame_length_ms=25.0,
        preemph_coeff=0.0, min_f0=50.0, max_f0=400.0, soft_min_f0=10.0,
        penalty_factor=0.1, lowpass_cutoff=1000.0, resample_freq=4000.0,
        delta_pitch=0.005, nccf_ballast=7000.0, lowpass_filter_width=1,
        upsample_filter_width=5, recompute_frame=500, snip_edges=1)


def default_pitch_post_options():
    """Kaldi ProcessPitchOptions defaults (reference pitch_kaldi.py:321-327)"""
    return PitchPostOptions(
        pitch_scale=2.0, pov_scale=2.0, pov_offset=0.0,
        delta_pitch_scale=10.0, delta_pitch_noise_stddev=0.005,
        normalization_left_context=75, normalization_right_context=75,
        delta_window=2, delay=0, add_pov_feature=1,
        add_normalized_log_pitch=1, add_delta_pitch=1, add_raw_log_pitch=0)


def default_options(kind):
    """An ``Options`` record with every Kaldi default filled in"""
    opts = Options()
    opts.kind = kind
    opts.frame = default_frame_options()
    opts.mel = default_mel_options()
    opts.use_energy = 1 if kind in (KIND_MFCC, KIND_PLP) else 0
    opts.energy_floor = 0.0
    opts.raw_energy = 1
    opts.htk_compat = 0
    opts.use_log_fbank = 1
    opts.use_power = 1
    opts.num_ceps = 13
    opts.cepstral_lifter = 22.0
    opts.lpc_order = 12
    opts.compress_factor = 1.0 / 3.0
    opts.cepstral_scale = 1.0
    opts.rasta = 0
    opts.compression = COMPRESSION['log']
    opts.delta_order = 2
    opts.delta_window = 2
    opts.append_deltas = 0
    opts.pitch = default_pitch_options()
    opts.pitch_post = default_pitch_post_options()
    opts.vad = VadOptions(
        energy_threshold=5.0, energy_mean_scale=0.5, frames_context=0,
        proportion_threshold=0.6)
    opts.sliding_cmvn = SlidingCmvnOptions(
        center=1, cmn_window=600, min_window=100, normalize_variance=0)
    opts.seed = 0
    return opts


def options_key(opts):
    """Hashable identity of an options record (plan cache key)"""
    return bytes(opts)


def copy_options(opts):
    """An independent copy of an options record (nested records included)"""
    return Options.from_buffer_copy(opts)
