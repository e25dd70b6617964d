#!/usr/bin/env python3
"""kernel_table.py <final_kernel_stats.csv> <pmc_traffic.json> <pmc_sq.json>: the per-kernel table of DESIGN.md section 3
("what bounds each kernel NOW"): microseconds per step from the rocprofv3 kernel trace, HBM-side bytes per launch from the two
--pmc passes, the rate they imply, matrix-pipe busy fraction and wait fraction from the SQ pass, as a markdown table."""
import csv, json, sys

LAYER = {  # template prefix -> layer role (batch 256, 128 x 128)
    "conv3x3_bwd_fused_limb_kernel<8, 8, 2,": "convt6 backward (+ convt7's data gradient gathered in the staging waves)",
    "conv3x3_bwd_fused_limb_kernel<8, 8, 1,": "conv2 backward",
    "conv3x3_bwd_fused_limb_kernel<16, 8, 0,": "convt5 backward",
    "conv3x3_bwd_fused_limb_kernel<8, 16, 0,": "conv3 backward",
    "conv3x3_bwd_fused_limb_kernel<16, 16, 1,": "conv4 backward",
    "conv3x3_bwd_fused_limb_kernel<16, 16, 2,": "convt4 backward",
    "conv3x3_bwd_fused_limb_kernel<16, 24, 0,": "conv5 backward",
    "conv3x3_bwd_fused_limb_kernel<24, 16, 0,": "convt3 backward",
    "conv3x3_bwd_fused_limb_kernel<24, 24, 1,": "conv6 backward",
    "conv3x3_bwd_fused_limb_kernel<24, 24, 2,": "convt2 backward",
    "conv3x3_bwd_fused_limb_kernel<24, 32, 0,": "conv7 backward",
    "conv3x3_bwd_fused_limb_kernel<32, 24, 0,": "convt1 backward",
    "thin_8to1_direct_fold_kernel": "convt7 forward + SSE + its weight gradient / BatchNorm-backward sums (round 5)",
    "thin_8to1_direct_kernel": "convt7 forward (eval / decode)",
    "thin_bwd_fused_1to8_kernel": "conv1 backward (y1 recomputed)",
    "thin_1to8_kernel": "conv1 forward",
    "up88_direct_kernel": "convt6 forward",
    "conv3x3_mfma_ws_kernel<8, 8, 1,": "conv2 forward", "conv3x3_mfma_ws_kernel<8, 16, 0,": "conv3 forward",
    "conv3x3_mfma_ws_kernel<16, 16, 1,": "conv4 forward", "conv3x3_mfma_ws_kernel<16, 24, 0,": "conv5 forward",
    "conv3x3_mfma_ws_kernel<24, 24, 1,": "conv6 forward", "conv3x3_mfma_kernel<24, 32, 0,": "conv7 forward (+ NCHW copy)",
    "conv3x3_mfma_ws_kernel<32, 24, 0,": "convt1 forward", "conv3x3_mfma_ws_kernel<24, 24, 2,": "convt2 forward",
    "conv3x3_mfma_ws_kernel<24, 16, 0,": "convt3 forward", "conv3x3_mfma_ws_kernel<16, 16, 2,": "convt4 forward",
    "conv3x3_mfma_ws_kernel<16, 8, 0,": "convt5 forward",
    "gemm_limb_kernel<128,": "fc1 / fc8 weight gradients (2 launches)", "gemm_limb_kernel<64, true, false": "fc8 forward, fc1 dX (2)",
    "gemm_limb_kernel<64, true, true": "fc1 forward, fc8 dX (2)", "adam_flat_kernel": "Adam",
    "wgrad_reduce_all_kernel": "weight-gradient partial rows -> gradient arena", "gemm_skinny_kernel<true, true>": "fc2, fc31|32|33, fc7 forward (3)",
    "gemm_skinny_kernel<true, false>": "their data gradients (3)", "gemm_skinny_grouped_kernel": "the eight small weight gradients",
    "nchw_to_nhwc_stats_kernel": "fc8 slabs -> NHWC + bn8 sums", "fc_mid_fwd_kernel": "heads, rsample, fc5, fc6", "fc_mid_bwd_kernel": "their backward",
    "splitk_reduce_kernel": "split-K reduce (2)", "pack_stats_kernel": "weight packs + input statistics + noise",
    "bn_bwd_apply_to_nchw_kernel": "bn8 backward -> NCHW", "relu_mask_to_nhwc_kernel": "fc1 dX slabs -> conv7's dU", "elbo_finalize_kernel": "ELBO",
}

def norm(n): return n.replace('void ', '').split('(')[0].strip()

def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    tr = {norm(k): v for k, v in json.load(open(sys.argv[2]))['kernels'].items()}
    sq = {norm(k): v for k, v in json.load(open(sys.argv[3]))['kernels'].items()}
    steps = max(int(r['Calls']) for r in rows if 'adam_flat' in r['Name'])
    out = []
    for r in rows:
        n = norm(r['Name'])
        if n.startswith('at::') or n.startswith('__amd') or 'bn_finalize' in n: continue
        calls = int(r['Calls']) / steps
        if calls < 0.5: continue
        avg = float(r['AverageNs']) / 1000
        t = next((v for k, v in tr.items() if k.startswith(n[:110]) or n.startswith(k[:110])), None)
        s = next((v for k, v in sq.items() if k.startswith(n[:80]) or n.startswith(k[:80])), None)
        mb = (t['read_bytes_per_step'] + t['write_bytes_per_step']) / t['launches_per_step'] / 1e6 if t else None
        role = next((v for k, v in LAYER.items() if n.startswith(k)), '')
        out.append((avg * calls, avg, calls, mb, s, n, role))
    out.sort(reverse=True)
    print("| µs / step | kernel | what | MB / launch | TB/s | MFMA busy | waiting |")
    print("|---|---|---|---|---|---|---|")
    tot = 0.0
    for us, avg, calls, mb, s, n, role in out:
        tot += us
        short = n if len(n) < 70 else n[:67] + '...'
        print("| %.1f%s | `%s` | %s | %s | %s | %s | %s |" % (
            us, '' if calls < 1.5 else ' (%d × %.1f)' % (round(calls), avg), short, role,
            '%.0f' % mb if mb is not None else '–', '%.2f' % (mb / avg) if mb else '–',
            '%.2f' % s['mfma_busy'] if s else '–', '%.2f' % s['wait_any'] if s else '–'))
    print("\nall kernels: %.1f µs per step" % tot)

if __name__ == '__main__':
    main()
