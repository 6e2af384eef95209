"""TEST INFRASTRUCTURE: an importable copy of the reference's PYTHON package under oracle/_ref/refpy/ (git-ignored like the
compiled reference next to it; it travels to the GPU box with the snapshot), so that tests can drive THIS repo's library through
the reference's own ctypes binding (pecos/core/base.py:799-976) and XLinearModel (pecos/xmc/xlinear/model.py).

The package is copied from /root/reference at build time (nothing of it enters the git history), the compiled reference
oracle/_ref/libpecos_float32.so is linked into pecos/core/, and the numpy-2 / scipy-1.15 shims of SURVEY.md 8(c) are applied
to pecos/utils/smat_util.py (the reference pins numpy<2, scipy<1.14)."""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def build(ref="/root/reference", dst=None):
    dst = dst or os.path.join(HERE, "_ref", "refpy")
    so = os.path.join(HERE, "_ref", "libpecos_float32.so")
    if not os.path.isdir(os.path.join(ref, "pecos")) or not os.path.exists(so):
        return None
    if os.path.exists(dst):
        shutil.rmtree(dst)
    os.makedirs(dst)
    shutil.copytree(os.path.join(ref, "pecos"), os.path.join(dst, "pecos"), ignore=shutil.ignore_patterns("*.hpp", "*.cpp", "*.h", "third_party", "__pycache__"))
    shutil.copy(so, os.path.join(dst, "pecos", "core", "libpecos_float32.so"))
    p = os.path.join(dst, "pecos", "utils", "smat_util.py")
    s = open(p).read()
    s = s.replace("smat.sputils.get_index_dtype", "smat._sputils.get_index_dtype")
    s = s.replace("smat.sputils.upcast", "smat._sputils.upcast")
    s = s.replace("np.array(X.indices, dtype=idx_dtype, copy=False)", "np.asarray(X.indices, dtype=idx_dtype)")
    s = s.replace("np.array(X.indptr, dtype=idx_dtype, copy=False)", "np.asarray(X.indptr, dtype=idx_dtype)")
    s = s.replace("copy=False)", "copy=None)")
    open(p, "w").write(s)
    return dst


if __name__ == "__main__":
    print(build(*(sys.argv[1:2])))
