"""Register / scratch / LDS budgets of the compiled kernels, read from the code objects inside pecos_amd/lib/libxrl_amd.so (no GPU needed).
Usage: python scripts/kernel_resources.py [name fragment ...] > profiles/rNN_kernel_resources.md   (default: the kernels the five workloads run)"""
import os
import pathlib
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
import test_kernel_resources as T  # noqa: E402

HOT = ["k1q_kernel<3, 0, false, true, false, false, false>", "k1q_kernel<3, 0, false, false, false, false, true>", "k1q_kernel<3, 0, false, true, false, true, false>",
       "k1q_kernel<3, 0, false, false, false, true, true>", "k1q_kernel<3, 0, false, true, false, false, true>", "k1q_kernel<1, 0, false, true, false, false, false>",
       "k1q_kernel<6, 0, false, false, false, false, false>", "k1q_kernel<16, 0, false, false, false, false, false>",
       "k1_kernel<32, 3, 0, false, 2>", "k1t_kernel<32, 3, 0, 2, false>", "k1t_kernel<32, 3, 0, 2, true>", "k1t_kernel<8, 1, 0, 0, true>", "k1_kernel<16, 1, 0, false, 0>", "k1_kernel<32, 1, 0, false, 0>", "k1g_kernel<2, 12, 1, 64, 0>", "k1g_kernel<2, 8, 1, 64, 0>",
       "k2_topk_wave<13>", "k2_topk_wave<2>", "k2_topk_reg", "tfidf_weight_kernel", "sort_scatter_kernel", "k0b_remaining"]


def waves(vgpr):
    # gfx950: 512 VGPRs per SIMD lane-slice, allocation granule 8, at most 8 wavefronts per SIMD
    g = (vgpr + 7) // 8 * 8
    return min(8, 512 // max(8, g))


def main():
    so = os.environ.get("PECOS_XRL_AMD_SO")     # another build of the library (kernel-tuning variants)
    frags = sys.argv[1:] or HOT
    notes = T.kernel_notes(pathlib.Path(tempfile.mkdtemp()), so)
    nice = T.demangle(sorted(notes))
    print("| kernel | VGPRs | wavefronts / SIMD | SGPRs | SGPR spills (to VGPR lanes) | VGPR spills | scratch B | static LDS B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for fr in frags:
        for k in sorted(notes):
            if fr in nice[k]:
                v = notes[k]
                print(f"| `{nice[k].replace('(anonymous namespace)::', '').replace('void xrl::', '').split('(')[0]}` | {v['vgpr']} | {waves(v['vgpr'])} | {v['sgpr']} | {v['sgpr_spill']} | {v['vgpr_spill']} | {v['scratch']} | {v['lds']} |")


if __name__ == "__main__":
    main()
