"""Process-wide knobs of pyramidkv_amd (read at call time)."""
import os

# How A / sqrt(head_dim) (reference pyramidkv_utils.py:317) is evaluated:
#   "div": fp32 division, what ATen does on CPU (the oracle's arithmetic)  - default
#   "rcp": multiply by the fp32 reciprocal, what ATen's GPU kernels do for a host-scalar divisor
scale_mode = os.environ.get("PKV_SCALE_MODE", "div")

# Read K/V of only the first head of each GQA group (the reference passes K/V already expanded by
# repeat_kv, llama_model.py:158-159, so the other heads of a group are byte-identical copies).
# Off by default: it is only valid when the caller really passes repeat_kv output.
gqa_dedup = os.environ.get("PKV_GQA_DEDUP", "0") == "1"

# Ada-SnapKV needs its head capacities on the host (klen_sum / max_seqlen_k are Python ints at the boundary).  1 (default):
# the budget kernel writes them into pinned host memory and the host polls a sequence word; 0: copy + stream synchronise.
host_poll = os.environ.get("PKV_HOST_POLL", "1") == "1"

# Ada-SnapKV, H * base <= 4096: every head's list of candidates starts at this many times the base budget instead of
# min(L, H * base) entries (exact: the kernel reports a list that ran out and the call is repeated with the full length);
# 0 = always the full length.  Needs host_poll.
ada_short_lists = int(os.environ.get("PKV_ADA_SHORT_LISTS", "4"))

# Ada-SnapKV prepared calls: host work that does not need the capacities (the outputs narrowed to the last call's total, the
# next call's metadata buffer) is done while the kernels run, when the capacities have not arrived yet (round 6).  0: off (A/B runs).
ada_prewait = os.environ.get("PKV_ADA_PREWAIT", "1") == "1"

# Order of equal scores in the selected rows: "canonical" = (value descending, index ascending), what PyTorch-ROCm's topk gives
# for k > 32; "aten_rocm" additionally reproduces, for k <= 32, the order its unstable small-slice sort leaves them in - the
# cache rows of PyramidKV's upper layers (k = 17..32) then match a reference run on the same GPU row for row.
tie_order = os.environ.get("PKV_TIE_ORDER", "canonical")
