"""C-ABI surface checks that run without a GPU: the library builds/loads, exports every symbol the header
declares, and the product refuses to compute on the CPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from arvae_amd import _lib, build
    build.build_library(verbose=False)
    return _lib.load()


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'arvae_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(arvae_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from arvae_amd import _lib
    names = header_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f'{name} declared in arvae_hip.h but not exported'
        assert name in _lib.SIGNATURES, f'{name} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == names


def test_ctypes_structs_match_the_header(tmp_path):
    """sizeof / field offsets of every struct in include/arvae_hip.h as gcc lays them out == the ctypes mirrors."""
    import shutil
    import subprocess
    from arvae_amd import _lib
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('gcc not available')
    structs = {'arvae_link_t': (_lib.LinkDesc, 'n', 'lo_perm_hw'), 'arvae_operand_t': (_lib.OperandDesc, 'v', 'act'),
               'arvae_gru_seq_t': (_lib.GruSeqDesc, 'gi', 'dh0_stride'),
               'arvae_tick_weights_t': (_lib.TickWeights, 'w_hh0', 'b_out'),
               'arvae_dense_wgrad_job_t': (_lib.DenseWgradJob, 'g', 'n_out'),
               'arvae_image_vae_t': (_lib.ImageVaeDesc, 'n_enc', 'flags'),
               'arvae_milestones_t': (_lib.Milestones, 'z_ready', 'linear_grads'),
               'arvae_measure_vae_t': (_lib.MeasureVaeDesc, 'vocab', 'rng_dev_step'),
               'arvae_measure_tables_t': (_lib.MeasureTables, 'midi_lut', 'rhythm_norm')}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/arvae_hip.h"', 'int main(void) {']
    for name, (_, first, last) in structs.items():
        lines.append(f'printf("{name} %zu %zu %zu\\n", sizeof({name}), offsetof({name}, {first}), offsetof({name}, {last}));')
    lines += ['return 0; }']
    src = tmp_path / 'sizes.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'sizes'
    subprocess.run([gcc, '-o', str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split('\n')
    for line in filter(None, out):
        name, size, off_first, off_last = line.split()
        cls, first, last = structs[name]
        assert ctypes.sizeof(cls) == int(size), name
        assert getattr(cls, first).offset == int(off_first) and getattr(cls, last).offset == int(off_last), name


def test_abi_version_and_error_string(lib):
    from arvae_amd import _lib
    assert lib.arvae_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.arvae_last_error_string(), bytes)


def test_argument_validation_without_gpu(lib):
    """bad arguments are rejected before any launch (pure host code)."""
    from arvae_amd._lib import LinkDesc, OperandDesc
    rc = lib.arvae_link_down(None, None, None, None, 0, None, None, None, None)
    assert rc == -1 and b'null' in lib.arvae_last_error_string()
    d = LinkDesc(1, 8, 8, 4, 5, 5, 4, 4, 4, 2, 1, 0, 0, 0, 0)       # lo extent should be 4x4
    op = OperandDesc(ctypes.c_void_p(16), None, None, 0)
    rc = lib.arvae_link_down(ctypes.byref(d), ctypes.byref(op), ctypes.c_void_p(16), None, 0, None,
                             ctypes.c_void_p(16), None, None)
    assert rc == -1 and b'does not match' in lib.arvae_last_error_string()
    # Morpho-MNIST 64 -> 64 k4 s1: the row-staged kernel's weights as two scaled fp16 terms in per-lane operand order + their
    # inverse scale, and the source's 1024 partial maxima (conv64s.hip)
    staged = 16 * 64 * 64 + 4 + 1024
    wide = LinkDesc(4, 25, 25, 64, 22, 22, 64, 4, 4, 1, 0, 0, 0, 0, 0)
    assert lib.arvae_link_ws_floats(ctypes.byref(wide)) == staged
    # 8 -> 64 channels (reduction side 8): the gathering kernel's re-ordered fp32 copy
    narrow = LinkDesc(4, 22, 22, 8, 19, 19, 64, 4, 4, 1, 0, 0, 0, 0, 0)
    assert lib.arvae_link_ws_floats(ctypes.byref(narrow)) in (16 * 8 * 64 + 4, staged)      # (+ the weights' maximum for the fp16 kernel)
    # 32 <-> 32 channels k4 s2 p1: the layer's weights as two scaled fp16 terms per value in per-lane operand order (Down and
    # Up parts, + the inverse scale), and the input's 1024 partial maxima
    assert lib.arvae_link_ws_floats(ctypes.byref(LinkDesc(4, 32, 32, 32, 16, 16, 32, 4, 4, 2, 1, 0, 0, 0, 0))) == (2 * 32 * 64 + 4 * 16 * 64 + 1) * 4 + 2 * 1024
    assert lib.arvae_reg_loss_ws_floats(512, 5) == 2 * 512 * 5
    assert lib.arvae_adam_step(None, None, None, None, 0, 0, 1e-4, 0.9, 0.999, 1e-8, 1.0, 0, None, None) == -1
    # the MeasureVAE sequence entry points
    from arvae_amd._lib import GruSeqDesc, TickWeights, DenseWgradJob
    assert lib.arvae_gru_seq_supported(128) == 1 and lib.arvae_gru_seq_supported(512) == 0
    assert lib.arvae_gru_seq_fwd(None, 1, 24, 16, 128, None) == -1
    seqs = (GruSeqDesc * 1)()
    assert lib.arvae_gru_seq_fwd(seqs, 1, 24, 16, 128, None) == -1 and b'null' in lib.arvae_last_error_string()
    assert lib.arvae_gru_seq_fwd(seqs, 1, 24, 16, 100, None) == -1 and b'hidden size' in lib.arvae_last_error_string()
    assert lib.arvae_gru_seq_bwd(seqs, 5, 24, 16, 128, None) == -1
    assert lib.arvae_tick_free_run(ctypes.byref(TickWeights()), None, None, 0, None, None, None, 2.0, 8, 4, 6, 128, 35, None,
                                   None, None) == -1
    assert lib.arvae_tick_free_run_ws_floats(128) == 3 * 3 * 128 * 128 * 3 // 2
    assert lib.arvae_embed_bwd_ws_floats(256, 24, 10, 35) == (256 * 24 // 64) * 35 * 10
    assert lib.arvae_dense_wgrad_batch((DenseWgradJob * 1)(), 1, None) == -1
    assert lib.arvae_operand_apply(None, 4, None, None) == -1


def test_cpu_tensors_are_refused():
    from arvae_amd import ops
    z = torch.randn(8, 4)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.reg_loss(z, torch.randn(8, 4), (1,), 1.0, 1.0)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.latent_head(z, z, z)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from arvae_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU'):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'ar-vae_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f'{f} imports the oracle'


def test_state_dict_keys_match_reference_layout():
    """same keys, order and shapes as the reference modules (recorded in oracle.image_vae.*_SHAPES)."""
    from arvae_amd.image_vae import DspritesVAE, MnistVAE
    from oracle import image_vae as o
    for cls, shapes, nparam in ((DspritesVAE, o.DSPRITES_SHAPES, 502005), (MnistVAE, o.MNIST_SHAPES, 1644145)):
        m = cls()
        sd = m.state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
        assert sum(p.numel() for p in m.parameters()) == nparam
        assert [k for k, _ in m.named_parameters()] == list(shapes.keys())


def test_comm_entry_points_without_a_gpu(lib):
    """the data-parallel collectives' ABI (arvae_comm_*): RCCL is found at run time (the copy torch already loaded), a unique
    id can be made on the host, and bad arguments / non-communicators are refused with a message -- no GPU is touched."""
    version = lib.arvae_comm_available()
    assert version > 20000, lib.arvae_last_error_string()                     # RCCL 2.x.y as 2xxyy
    a, b = (ctypes.c_char * 128)(), (ctypes.c_char * 128)()
    assert lib.arvae_comm_unique_id(a) == 0 and lib.arvae_comm_unique_id(b) == 0
    assert bytes(a) != bytes(b) and bytes(a) != bytes(128)
    assert lib.arvae_comm_unique_id(None) == -1
    handle = ctypes.c_void_p()
    assert lib.arvae_comm_init(bytes(a), 2, 2, 1000, ctypes.byref(handle)) == -1     # rank out of range: refused before RCCL is called
    assert b'rank 2 of 2' in lib.arvae_last_error_string()
    assert lib.arvae_comm_init(None, 0, 1, 1000, ctypes.byref(handle)) == -1
    fake = ctypes.create_string_buffer(64)                                     # not a communicator: the magic word is missing
    for call in (lambda: lib.arvae_comm_destroy(fake), lambda: lib.arvae_comm_rank(fake), lambda: lib.arvae_comm_world(fake),
                 lambda: lib.arvae_comm_async_error(fake),
                 lambda: lib.arvae_comm_all_reduce(fake, None, 0, 0, 0, None),
                 lambda: lib.arvae_comm_all_gather(fake, None, None, 0, 0, None),
                 lambda: lib.arvae_comm_broadcast(fake, None, 0, 0, 0, None)):
        assert call() == -1


def test_library_transport_needs_a_gpu():
    """parallel.LibraryComm is the HIP path's transport: no CPU stand-in"""
    if torch.cuda.is_available():
        pytest.skip('CPU-only check')
    from arvae_amd import parallel
    with pytest.raises(RuntimeError, match='needs a GPU'):
        parallel.LibraryComm(0, 1)


def test_product_library_reads_no_environment(lib):
    """the product library has no run-time switches (csrc/diag.h): it does not even import getenv; the diagnostic build of the
    same sources does, and both export the same ABI"""
    import shutil
    import subprocess
    from arvae_amd import build
    nm = shutil.which('nm')
    if nm is None:
        pytest.skip('nm not available')

    def undefined(path):
        out = subprocess.run([nm, '-D', '--undefined-only', path], check=True, capture_output=True, text=True).stdout
        return {ln.split()[-1].split('@')[0] for ln in out.splitlines() if ln.strip()}

    def exported(path):
        out = subprocess.run([nm, '-D', '--defined-only', path], check=True, capture_output=True, text=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if ' T ' in ln and 'arvae_' in ln}
    assert 'getenv' not in undefined(build.LIB_PATH) and 'secure_getenv' not in undefined(build.LIB_PATH)
    assert 'getenv' in undefined(build.DIAG_LIB_PATH)
    # (the diagnostic build may add arvae_debug_* readers for its instrumentation; the ABI proper is the same)
    assert exported(build.LIB_PATH) == {s for s in exported(build.DIAG_LIB_PATH) if 'arvae_debug_' not in s}


def test_measure_executor_descriptor_without_gpu(lib):
    """the whole-model MeasureVAE executor's host side on a CPU-only box: the descriptor FusedMeasureVAE builds over a FlatAdam arena
    (offsets inside the arena, the three pairs of tensors back to back), the workspace size the library reports for it, and the
    argument checks of the three entry points (pure host code: nothing is launched)."""
    import torch
    from arvae_amd import synthetic as syn
    from arvae_amd.fused_measure import FusedMeasureVAE
    from arvae_amd.measure_vae import MeasureVAE
    from arvae_amd.optim import FlatAdam

    class Folk:
        class_name = '4by4_FolkNBarDataset_1_'
        n_bars = 1

        def __init__(self):
            self.index2note_dicts, self.note2index_dicts = syn.measure_vocabulary()

        def __repr__(self):
            return self.class_name
    torch.manual_seed(0)
    model = MeasureVAE(Folk(), 10, 2, 2, 128, 0.5, 32, 2, 128, 0.5, False, 'folk')
    opt = FlatAdam(model.arena_parameters(), lr=1e-4)
    fused = FusedMeasureVAE(model, opt, (0, 1, 2, 3), 0.001, 1.0, 10.0)
    d = fused.descriptor()
    total = opt.param_arena.numel()
    offs = [d.enc_table, d.enc_w_ih[0], d.enc_w_ih[1], d.enc_b_ih[0], d.enc_w_hh[1][1], d.head_w0, d.mean_w2, d.dec_table, d.x0, d.b0, d.z2beat_w,
            d.beat_w_ih[0], d.tick_init_w, d.tick_w_ih[0], d.tick_w_hh[1], d.out_w, d.out_b]
    assert all(0 <= o < total and o % 4 == 0 for o in offs) and len(set(offs)) == len(offs)
    # the reverse direction's projection follows the forward direction's; the paired heads / tick-initialisation layers likewise
    gru = model.encoder.lstm
    assert d.enc_w_ih[0] + gru.weight_ih_l0.numel() == fused._offset(gru.weight_ih_l0_reverse)
    assert d.head_w0 + model.encoder.linear_mean[0].weight.numel() == fused._offset(model.encoder.linear_log_std[0].weight)
    assert d.tick_init_w + model.decoder.beat_emb_to_tick_rnn_hidden[0].weight.numel() == fused._offset(model.decoder.beat_emb_to_tick_rnn_input[0].weight)
    assert (d.vocab, d.emb, d.enc_hidden, d.dec_hidden, d.zdim, d.steps, d.beats, d.ticks_per_beat, d.n_reg) == (35, 10, 128, 128, 32, 24, 4, 6, 4)
    ws256, ws8 = lib.arvae_measure_vae_ws_floats(ctypes.byref(d), 256), lib.arvae_measure_vae_ws_floats(ctypes.byref(d), 8)
    assert ws256 > ws8 > 0 and ws256 % 4 == 0 and ws256 * 4 < 2 ** 31          # (some 400 MB at the benchmark batch)
    assert lib.arvae_measure_vae_ws_floats(ctypes.byref(d), 0) == -1 and lib.arvae_measure_vae_ws_floats(None, 8) == -1
    assert lib.arvae_measure_vae_ws_floats(ctypes.byref(d), 4096) == -1 and b'segment sums' in lib.arvae_last_error_string()
    assert lib.arvae_measure_vae_forward(ctypes.byref(d), 8, None, None, None, None, None, 1, None, None, None, None, None, None, None, None,
                                         None, 0, None) == -1 and b'null pointer' in lib.arvae_last_error_string()
    assert lib.arvae_measure_vae_finish(ctypes.byref(d), 8, None, None, None, 16, 2.0, None, None, None, None, None, None, None) == -1
    assert lib.arvae_measure_vae_backward(ctypes.byref(d), 8, None, None, None, None, None, None, None, None, None, None, None, None, None,
                                          1.0, None, None) == -1
    d.steps = 23
    assert lib.arvae_measure_vae_ws_floats(ctypes.byref(d), 8) == -1 and b'beats * ticks_per_beat' in lib.arvae_last_error_string()
    # torch's parameters() order does not put the pairs side by side: the trainer then keeps the per-layer path
    other = FlatAdam(model.parameters(), lr=1e-4)
    assert 'adjacent' in FusedMeasureVAE.supports(model, other, ())
