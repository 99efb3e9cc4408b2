"""Prefill wall time of a Llama-3-8B-shaped stack (2 layers, random weights) through the transformers adapter, PyramidKV
budget 128 - what the eviction step weighs inside a layer.  Measured with update_kv on the main stream ("sync") and, in an
experiment that was removed again, on a side HIP stream joined at the first decode step ("async", config.async_compact):
S = 32768: 61.2-61.8 ms vs 61.1-61.5 ms for the two layers, S = 8192: 9.9-10.1 vs 9.8-9.9 ms - the eviction (0.04 ms per
layer) is 0.14 % of a layer's prefill, and hiding it is not measurable.  Without the removed option this script times the
synchronous order only."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyramidkv_amd as P
from pyramidkv_amd import monkeypatch as mp, config as cfgmod
from transformers import LlamaConfig, LlamaForCausalLM, DynamicCache
torch.manual_seed(0)
cfg = LlamaConfig(vocab_size=1024, hidden_size=4096, intermediate_size=14336, num_hidden_layers=2, num_attention_heads=32,
                  num_key_value_heads=8, head_dim=128, max_position_embeddings=65536)
model = LlamaForCausalLM(cfg).to(torch.bfloat16).cuda().eval()
res = {}
mp.replace_llama("pyramidkv")
for layer in model.model.layers:
    c = layer.self_attn.config
    c.window_size, c.max_capacity_prompt, c.kernel_size, c.pooling, c.merge = 8, 128, 7, "maxpool", None
for S in (8192, 32768):
    ids = torch.randint(0, 1024, (1, S), device="cuda")
    for mode in (False, False):
        ts = []
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                model(ids, past_key_values=DynamicCache(config=cfg), use_cache=True, logits_to_keep=1)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res.setdefault("S%d_%s" % (S, "async" if mode else "sync"), []).append(round(1e3 * min(ts[1:]), 3))
print(json.dumps(res))
