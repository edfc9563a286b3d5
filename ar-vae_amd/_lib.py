"""ctypes binding of libarvae_hip.so (see include/arvae_hip.h).

There is NO fallback: if the library is missing or a call is made without a GPU
the caller gets a RuntimeError.  Build with ``python __graft_entry__.py`` or
``python ar-vae_amd/build.py``.
"""
import ctypes
import os
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, 'libarvae_hip.so')
ABI_VERSION = 11

c_i32, c_i64, c_f32, c_f64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p


class LinkDesc(ctypes.Structure):
    """arvae_link_t"""
    _fields_ = [(n, c_i32) for n in ('n', 'hh', 'hw', 'chi', 'lh', 'lw', 'clo', 'kh', 'kw', 'stride', 'pad',
                                      'hi_perm_c', 'hi_perm_hw', 'lo_perm_c', 'lo_perm_hw')]


class OperandDesc(ctypes.Structure):
    """arvae_operand_t"""
    _fields_ = [('v', c_vp), ('y', c_vp), ('mask', c_vp), ('act', c_i32)]


class LayerDesc(ctypes.Structure):
    """arvae_layer_t"""
    _fields_ = [('link', LinkDesc), ('is_up', c_i32), ('act', c_i32), ('dropout', c_i32), ('reserved', c_i32),
                ('w_off', c_i64), ('b_off', c_i64)]


MAX_LAYERS = 8


class ImageVaeDesc(ctypes.Structure):
    """arvae_image_vae_t"""
    _fields_ = [('n_enc', c_i32), ('n_dec', c_i32), ('enc', LayerDesc * MAX_LAYERS), ('dec', LayerDesc * MAX_LAYERS),
                ('head_mu', LayerDesc), ('head_log_std', LayerDesc), ('zdim', c_i32), ('recon_dist', c_i32),
                ('n_reg', c_i32), ('reg_dims', c_i32 * 16), ('beta', c_f32), ('gamma', c_f32), ('delta', c_f32),
                ('rng_eps', c_i32), ('rng_offset', ctypes.c_uint32), ('rng_step', ctypes.c_uint32), ('rng_seed', ctypes.c_uint64),
                ('rng_dev_step', c_vp), ('milestones', c_vp), ('status', c_vp), ('flags', c_i32), ('reserved', c_i32)]


class Milestones(ctypes.Structure):
    """arvae_milestones_t: hipEvent_t handles the whole-model executors record (data-parallel overlap)"""
    _fields_ = [('z_ready', c_vp), ('dec_grads', c_vp), ('linear_grads', c_vp)]


class GruSeqDesc(ctypes.Structure):
    """arvae_gru_seq_t"""
    _fields_ = [('gi', c_vp), ('gi_tstride', c_i64), ('w_hh', c_vp), ('b_hh', c_vp), ('h0', c_vp), ('h_all', c_vp),
                ('h_stride', c_i64), ('saved', c_vp), ('reverse', c_i32), ('reserved', c_i32), ('dh_all', c_vp),
                ('dh_stride', c_i64), ('dgi', c_vp), ('dgh', c_vp), ('dh0', c_vp), ('dh_last', c_vp),
                ('dh_last_stride', c_i64), ('h_prev_out', c_vp), ('gi_rstride', c_i64), ('dgi_rstride', c_i64), ('h_fin', c_vp),
                ('h_fin_stride', c_i64), ('h0_stride', c_i64), ('dh0_stride', c_i64)]


class TickWeights(ctypes.Structure):
    """arvae_tick_weights_t"""
    _fields_ = [(n, c_vp) for n in ('w_hh0', 'b_hh0', 'w_ih1', 'b_ih1', 'w_hh1', 'b_hh1', 'w_out', 'b_out')]


class MeasureVaeDesc(ctypes.Structure):
    """arvae_measure_vae_t"""
    _fields_ = ([(n, c_i32) for n in ('vocab', 'emb', 'enc_hidden', 'dec_hidden', 'zdim', 'steps', 'beats', 'ticks_per_beat')] +
                [('enc_table', c_i64), ('enc_w_ih', c_i64 * 2), ('enc_b_ih', c_i64 * 2), ('enc_w_hh', (c_i64 * 2) * 2),
                 ('enc_b_hh', (c_i64 * 2) * 2), ('head_w0', c_i64), ('head_b0', c_i64), ('mean_w2', c_i64), ('mean_b2', c_i64),
                 ('lstd_w2', c_i64), ('lstd_b2', c_i64), ('dec_table', c_i64), ('x0', c_i64), ('b0', c_i64), ('z2beat_w', c_i64),
                 ('z2beat_b', c_i64), ('beat_w_ih', c_i64 * 2), ('beat_b_ih', c_i64 * 2), ('beat_w_hh', c_i64 * 2),
                 ('beat_b_hh', c_i64 * 2), ('tick_init_w', c_i64), ('tick_init_b', c_i64), ('tick_w_ih', c_i64 * 2),
                 ('tick_b_ih', c_i64 * 2), ('tick_w_hh', c_i64 * 2), ('tick_b_hh', c_i64 * 2), ('out_w', c_i64), ('out_b', c_i64),
                 ('enc_dropout', c_f32), ('dec_dropout', c_f32), ('n_reg', c_i32), ('reg_dims', c_i32 * 16), ('beta', c_f32),
                 ('gamma', c_f32), ('delta', c_f32), ('rng_draw', c_i32), ('rng_offset', ctypes.c_uint32 * 3),
                 ('rng_step', ctypes.c_uint32), ('rng_seed', ctypes.c_uint64), ('rng_dev_step', c_vp)])


class MeasureTables(ctypes.Structure):
    """arvae_measure_tables_t"""
    _fields_ = [('midi_lut', c_vp), ('is_note', c_vp), ('is_density_note', c_vp), ('rhythm_weights', c_vp), ('rhythm_norm', c_f32)]


class DenseWgradJob(ctypes.Structure):
    """arvae_dense_wgrad_job_t"""
    _fields_ = [('g', OperandDesc), ('x', c_vp), ('dw', c_vp), ('dbias', c_vp), ('rows', c_i32), ('n_in', c_i32),
                ('n_out', c_i32), ('reserved', c_i32)]


_P = ctypes.POINTER
# name -> (restype, argtypes); must list every symbol declared in include/arvae_hip.h
SIGNATURES = {
    'arvae_abi_version': (c_i32, []),
    'arvae_last_error_string': (ctypes.c_char_p, []),
    'arvae_device_count': (c_i32, []),
    'arvae_profile_begin': (c_i32, [c_vp]),
    'arvae_profile_end': (c_i64, [ctypes.c_char_p, c_i64]),
    'arvae_link_ws_floats': (c_i64, [_P(LinkDesc)]),
    'arvae_link_down': (c_i32, [_P(LinkDesc), _P(OperandDesc), c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'arvae_link_up': (c_i32, [_P(LinkDesc), _P(OperandDesc), c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'arvae_link_wgrad_ws_floats': (c_i64, [_P(LinkDesc)]),
    'arvae_link_wgrad': (c_i32, [_P(LinkDesc), _P(OperandDesc), _P(OperandDesc), c_vp, c_vp, c_i32, c_vp, c_vp]),
    'arvae_channel_sum_ws_floats': (c_i64, [c_i64, c_i32]),
    'arvae_channel_sum': (c_i32, [_P(OperandDesc), c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'arvae_dense_wgrad_batch': (c_i32, [_P(DenseWgradJob), c_i32, c_vp]),
    'arvae_operand_apply': (c_i32, [_P(OperandDesc), c_i64, c_vp, c_vp]),
    'arvae_latent_fwd': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'arvae_latent_bwd': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    'arvae_kld_fwd': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp]),
    'arvae_kld_bwd': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'arvae_reg_loss_ws_floats': (c_i64, [c_i64, c_i32]),
    'arvae_reg_loss': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i64, _P(c_i32), c_i32, c_f32, c_f32,
                               c_vp, c_vp, c_vp, c_vp]),
    'arvae_recon_ws_floats': (c_i64, [c_i64]),
    'arvae_image_recon': (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'arvae_token_recon': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'arvae_scale_by_scalar': (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    'arvae_gru_gates_fwd': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'arvae_gru_gates_bwd': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'arvae_gru_seq_supported': (c_i32, [c_i32]),
    'arvae_gru_seq_fwd': (c_i32, [_P(GruSeqDesc), c_i32, c_i32, c_i32, c_i32, c_vp]),
    'arvae_gru_seq_bwd': (c_i32, [_P(GruSeqDesc), c_i32, c_i32, c_i32, c_i32, c_vp]),
    'arvae_tick_free_run_ws_floats': (c_i64, [c_i32]),
    'arvae_tick_free_run_supported': (c_i32, [c_i32, c_i32]),
    'arvae_tick_free_run': (c_i32, [_P(TickWeights), c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                    c_vp, c_vp, c_vp]),
    'arvae_embed_fwd': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'arvae_embed_bwd_ws_floats': (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    'arvae_embed_bwd': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    'arvae_tick_rows_fwd': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'arvae_tick_rows_bwd': (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'arvae_tick_gi_fwd': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'arvae_tick_gi_bwd_ws_floats': (c_i64, [c_i32, c_i32]),
    'arvae_tick_gi_bwd': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    'arvae_row_argmax': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    'arvae_concat_cols': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    'arvae_split_cols': (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp]),
    'arvae_scale_mask': (c_i32, [c_vp, c_vp, c_f32, c_i64, c_i32, c_vp, c_vp]),
    'arvae_broadcast_rows': (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    'arvae_gather_rows_u8': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_f32, c_vp, c_vp]),
    'arvae_measure_attributes': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_f32, c_vp, c_vp]),
    'arvae_image_vae_ws_floats': (c_i64, [_P(ImageVaeDesc), c_i32, c_i64]),
    'arvae_image_vae_forward': (c_i32, [_P(ImageVaeDesc), c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, _P(c_vp), c_vp, c_vp,
                                        c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'arvae_image_vae_finish': (c_i32, [_P(ImageVaeDesc), c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_vp]),
    'arvae_image_vae_backward': (c_i32, [_P(ImageVaeDesc), c_i32, c_vp, c_vp, c_vp, c_vp, _P(c_vp), c_vp, c_vp, c_vp,
                                         c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_vp, c_vp]),
    'arvae_measure_vae_ws_floats': (c_i64, [_P(MeasureVaeDesc), c_i32]),
    'arvae_measure_vae_forward': (c_i32, [_P(MeasureVaeDesc), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, _P(MeasureTables), c_vp,
                                          c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'arvae_measure_vae_finish': (c_i32, [_P(MeasureVaeDesc), c_i32, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                         c_vp]),
    'arvae_measure_vae_backward': (c_i32, [_P(MeasureVaeDesc), c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_vp, c_f32, c_vp, c_vp]),
    'arvae_philox_normal': (c_i32, [c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp]),
    'arvae_philox_keep_mask': (c_i32, [c_vp, c_i64, c_f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp]),
    'arvae_philox_keep_masks': (c_i32, [c_i32, c_vp, c_vp, c_f32, ctypes.c_uint64, c_vp, ctypes.c_uint32, c_vp, c_vp]),
    'arvae_count_nonfinite': (c_i32, [c_vp, c_i64, c_vp, c_vp]),
    'arvae_count_out_of_range': (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    'arvae_adam_step': (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_f64, c_f64, c_f64, c_f64, c_f32, c_i32, c_vp, c_vp]),
    'arvae_comm_available': (c_i32, []),
    'arvae_comm_unique_id': (c_i32, [c_vp]),
    'arvae_comm_init': (c_i32, [c_vp, c_i32, c_i32, c_i32, _P(c_vp)]),
    'arvae_comm_destroy': (c_i32, [c_vp]),
    'arvae_comm_abort': (c_i32, [c_vp]),
    'arvae_comm_rank': (c_i32, [c_vp]),
    'arvae_comm_world': (c_i32, [c_vp]),
    'arvae_comm_async_error': (c_i32, [c_vp]),
    'arvae_comm_all_gather': (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    'arvae_comm_all_reduce': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    'arvae_comm_broadcast': (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    'arvae_comm_group_begin': (c_i32, []),
    'arvae_comm_group_end': (c_i32, []),
}

_lock = threading.Lock()
_lib = None


def load():
    """dlopen the library and bind every symbol; raises RuntimeError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = os.environ.get('ARVAE_LIB') or LIB_PATH      # ARVAE_LIB: a diagnostic build of the same sources (tools/bin/*.so)
        if not os.path.exists(path):
            raise RuntimeError(
                f'{path} is missing: the AR-VAE HIP kernels are not built and there is no CPU '
                f'fallback. Run `python __graft_entry__.py` (or `python ar-vae_amd/build.py`) first.')
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).
        # It must be the ONE runtime in the process: import torch first so that our NEEDED
        # libamdhip64.so.7 resolves to the copy torch already loaded.  Loading this library first would
        # pull in /opt/rocm's runtime as a second instance, which then sees no device.
        import torch  # noqa: F401
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        got = lib.arvae_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f'libarvae_hip.so ABI {got} != expected {ABI_VERSION}: rebuild it')
        _lib = lib
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = load().arvae_last_error_string().decode(errors='replace')
        raise RuntimeError(f'libarvae_hip {what} failed ({rc}): {msg}')
