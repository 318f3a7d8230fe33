#!/usr/bin/env python
"""Copy the judged summaries of tools/gpu_r3_prof_b.sh (gpurun_out/prof_r03*/summary.txt, gpurun_out/r3prof/*.txt) into
profiles/ (tracked), with a header saying what was run.   python tools/collect_r3_profiles.py"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def keep(line):
    return not any(t in line for t in ("at::native", "rocclr", "at::cuda", "amdgpu.ids")) and not line.startswith(("W2026", "void  "))


def copy(src, dst, header, width=420):
    lines = [l.rstrip("\n")[:width] for l in open(os.path.join(G, src)) if keep(l)]
    with open(os.path.join(P, dst), "w") as f:
        f.write("".join("# " + h + "\n" for h in header))
        f.write("\n".join(lines) + "\n")
    print(dst, len(lines), "lines")


def split_summary(src):
    """summary.txt of tools/summarize_prof.py -> (trace section, pmc sections)"""
    lines = [l for l in open(os.path.join(G, src)) if keep(l)]
    cut = next((i for i, l in enumerate(lines) if l.startswith("== pmc")), len(lines))
    return lines[:cut], lines[cut:]


tr, pmc = split_summary("prof_r03t/summary.txt")
with open(os.path.join(P, "r03_train_step_kernel_trace_summary.txt"), "w") as f:
    f.write("# bash tools/gpu_pmc_train.sh r03t: rocprofv3 --kernel-trace --stats -- python3 tools/prof_train.py 4  (NRMS grad step, B=64, H=25, C=5,\n"
            "# S=50, D=768; 7 steps incl. warm-up), round-3 binary, MI355X.  ' grid=129024x1' = gemm_dw256_kernel (its name starts with an\n"
            "# anonymous namespace the summariser drops), ' grid=196608x1' / 92160 / 124416 = gemm_dw_kernel (live-row launches).\n")
    f.writelines(l[:200] + ("\n" if not l.endswith("\n") else "") if len(l) > 200 else l for l in tr)
with open(os.path.join(P, "r03_train_step_pmc.txt"), "w") as f:
    f.write("# bash tools/gpu_pmc_train.sh r03t: one rocprofv3 --pmc pass per counter group (kernel trace only) of python3 tools/prof_train.py 2,\n"
            "# mean per dispatch.  MFMA busy fraction of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).\n")
    f.writelines(l[:420] + "\n" if len(l) > 420 else l for l in pmc)
print("train", len(tr), len(pmc))
for m, what in (("standard", "StandardRec"), ("NAML", "NAML")):
    copy(f"r3prof/trace_{m}.txt", f"r03_{m.lower()}_fwd_kernel_trace_after.txt" if m == "NAML" else "r03_standardrec_fwd_kernel_trace_after.txt",
         [f"bash tools/gpu_trace.sh r03_{m} tools/prof_other_models.py {m} 6: rocprofv3 --kernel-trace --stats of 6 forward passes of {what}",
          "(B=512, H=25, C=5, S=50, D=768), round-3 binary with additive_fused_kernel (DESIGN.md section 4.7); the *_before files are the",
          "same command on the two-launch pipeline at the start of the round"], 200)
    copy(f"r3prof/pmc_{m}.txt", f"r03_{'naml' if m == 'NAML' else 'standardrec'}_fwd_pmc.txt",
         [f"PMC_SETS='sq lds' bash tools/gpu_pmc_any.sh r03_{m} prof_other_models.py {m} 4: separate rocprofv3 --pmc passes (kernel trace only), mean per dispatch"])
copy("r3prof/pmc_nf.txt", "r03_news_fused_pmc.txt",
     ["PMC_SETS='sq lds' bash tools/gpu_pmc_any.sh r03_nf prof_news.py 1024 8 30 320 16: the fused short-title kernel with the fold inside (DESIGN.md 4.4),",
      "1 024 news x 30 x 320 / 16 heads; LDS bank conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"])
copy("r3prof/bench_af.txt", "r03_additive_fused.txt",
     ["python tools/bench_af.py: the one-launch additive encoder against GEMM + pooling and against plain GEMMs of the same kernel family,",
      "settled clocks (0.6 s of back-to-back calls before every timing); last column = fraction of the 157.3 TFLOP/s fp32 matrix peak"])
copy("r3prof/bench_dw.txt", "r03_gemm_dw.txt",
     ["python tools/bench_dw.py: dW[768,768] = dY^T . X over 80 000 rows through xnrs_linear_bwd, one line per kernel choice (interleaved, settled clocks)"])
with open(os.path.join(P, "r03_news_fused_dispatch_sweep.txt"), "w") as f:
    f.write("# python tools/bench_news_fused.py and NF_SWEEP=1 python tools/bench_news_fused.py: fused short-title kernel (fold inside) end to end,\n"
            "# the kernel alone, one news per workgroup, and the per-token out-projection variant; executed-FLOP and reference-order TFLOP/s\n")
    for src in ("r3prof/bench_nf.txt", "r3prof/bench_nf_sweep.txt"):
        f.writelines(l for l in open(os.path.join(G, src)) if keep(l))
copy("r3prof/bench_compact.txt", "r03_device_compaction.txt",
     ["python tools/bench_compact.py: 25 600 news x 50 x 768 through the NRMS news encoder against the share of all-masked news:",
      "dense call, device-compacted (xnrs_text_encoder_fwd_compact, no host read), host-compacted (xnrs_text_encoder_fwd_unpadded, one nonzero)"])
