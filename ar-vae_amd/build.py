"""Builds libarvae_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

Two libraries from the same sources:
  libarvae_hip.so        the product: reads no environment variable (csrc/diag.h), one code path per kernel family;
  libarvae_hip_diag.so   -DARVAE_DIAG: the same code with its diagnostic switches live (ARVAE_NO_PAIR32, ARVAE_MIDBLOCK=0, ...);
                         `ARVAE_LIB=.../libarvae_hip_diag.so` selects it (same-box A/B runs, and the two parity tests that hold the
                         default paths to their alternatives)."""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libarvae_hip.so')
DIAG_LIB_PATH = os.path.join(PKG_DIR, 'libarvae_hip_diag.so')
ARCH = 'gfx950'
FLAGS = [f'--offload-arch={ARCH}', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']
FLAGS += os.environ.get('ARVAE_HIPCC_FLAGS', '').split()      # ablation / diagnostic builds only


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: libarvae_hip.so cannot be built')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, diag=False):
    """Compile every csrc/*.hip for gfx950 and link ar-vae_amd/libarvae_hip.so (and, with diag, libarvae_hip_diag.so: the
    diagnostic twin is built only on request -- `python ar-vae_amd/build.py --diag`, __graft_entry__.build(), tools/)."""
    path = _build(force, verbose, LIB_PATH, 'build', [])
    if diag:
        _build(force, verbose, DIAG_LIB_PATH, 'build_diag', ['-DARVAE_DIAG'])
    return path


def _build(force, verbose, lib_path, objdir_name, extra_flags):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, objdir_name)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(PKG_DIR), 'include', 'arvae_hip.h'))
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + extra_flags + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        if verbose and out.strip():
            print(out)
    if force or procs or _stale(lib_path, objs):
        cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', lib_path] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib_path


if __name__ == '__main__':
    build_library(force='--force' in sys.argv, diag='--diag' in sys.argv or '--force' in sys.argv)
