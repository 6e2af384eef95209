"""Static guard on the compiled kernels (no GPU): register budgets and scratch use of the hot kernels, read from the code objects inside
pecos_amd/lib/libxrl_amd.so (amdhsa metadata notes).  The query-stationary kernel is tuned to 64 VGPRs = 8 wavefronts per SIMD (DESIGN.md
section 4); an edit that pushes it past that, or makes the default instantiation spill to scratch, fails here instead of costing a GPU run."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import REPO

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_notes(tmp_path, so=None):
    so = so or os.path.join(REPO, "pecos_amd", "lib", "libxrl_amd.so")
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(so) and os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("library or llvm tools not present")
    work = str(tmp_path / "co")
    os.makedirs(work)
    shutil.copy(so, work)                                   # --offloading writes the bundles next to its input
    subprocess.check_call([objdump, "--offloading", os.path.join(work, os.path.basename(so))], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(work)):
        if "amdgcn" not in f:
            continue
        txt = subprocess.check_output([readelf, "--notes", os.path.join(work, f)], stderr=subprocess.DEVNULL).decode()
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))          # noqa: E731
            out[name.group(1)] = dict(vgpr=get("vgpr_count"), sgpr=get("sgpr_count"), scratch=get("private_segment_fixed_size"),
                                      vgpr_spill=get("vgpr_spill_count"), sgpr_spill=get("sgpr_spill_count"), lds=get("group_segment_fixed_size"))
    return out


def demangle(names):
    filt = next((f for f in (os.path.join(LLVM, "llvm-cxxfilt"), shutil.which("c++filt") or "") if f and os.path.exists(f)), None)
    if filt is None:
        pytest.skip("no C++ demangler here")
    res = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, res))


def test_hot_kernels_keep_their_register_budget(tmp_path):
    notes = kernel_notes(tmp_path)
    assert len(notes) > 50, len(notes)
    nice = demangle(sorted(notes))
    by = {nice[k]: v for k, v in notes.items()}
    find = lambda frag: {k: v for k, v in by.items() if frag in k}                       # noqa: E731
    # K1Q <NSMAX, PPC, DENSEX, MULTI, BIASF, PRES, BIGW>: the sparse instantiations the Amazon / Eurlex / Wiki10 workloads run -- the fused
    # narrow levels (matrices < 4 GiB) and the sorted launch of Amazon-670K's level 3 (4.4 GiB: BIGW)
    for frag in ("k1q_kernel<3, 0, false, true, false, false, false>", "k1q_kernel<3, 0, false, false, false, false, true>"):
        ks = find(frag)
        assert len(ks) == 1, frag
        d = next(iter(ks.values()))
        assert d["vgpr"] <= 64 and d["scratch"] <= (0 if ", true, false, false, false>" in frag else 64) and d["vgpr_spill"] == 0, (frag, d)  # 8 wavefronts per SIMD; the fused kernel keeps
                                                                                       # everything in registers, the single-layer one parks a few SGPRs in scratch outside its loops (80-SGPR budget at 8 wavefronts)
    # the presence-word variants (layers that run unstaged: the hard workload): 8 wavefronts; spills outside the feature loops are tolerated, bounded
    for frag in ("k1q_kernel<3, 0, false, false, false, true, true>", "k1q_kernel<3, 0, false, true, false, true, false>"):
        pres = next(iter(find(frag).values()))
        assert pres["vgpr"] <= 64 and pres["scratch"] <= 256, (frag, pres)
    # the tile kernel of the leaf and the dense-query SGEMM: no scratch
    for frag in ("k1_kernel<32, 3, 0, false, 2>", "k1_kernel<16, 1, 0, false, 0>"):
        ks = find(frag)
        assert ks and all(v["scratch"] == 0 and v["vgpr_spill"] == 0 for v in ks.values()), (frag, ks)
        assert all(v["vgpr"] <= 96 for v in ks.values()), (frag, ks)                     # amdgpu_waves_per_eu(5, 8): >= 5 wavefronts
    k1g = find("k1g_")
    assert k1g and all(v["scratch"] == 0 for v in k1g.values()), k1g
    topk = find("k2_topk_wave<13>")
    assert topk and all(v["scratch"] == 0 and v["vgpr"] <= 64 for v in topk.values()), topk
    tf = find("tfidf_weight_kernel")
    assert tf and all(v["scratch"] == 0 for v in tf.values())
