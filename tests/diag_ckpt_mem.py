"""Diagnostic (test infrastructure - it uses the oracle's seeded initialisers, so it lives under tests/; not collected by pytest): which saved
activations survive a checkpointed forward (memory after forward / peak, weakrefs of everything _save saw).   python tests/diag_ckpt_mem.py"""
import gc
import os
import sys
import weakref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llama-x_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from llx import ops  # noqa: E402
from oracle import ref as O  # noqa: E402
from tests.util import bf16_params, build_model  # noqa: E402

cuda = torch.device("cuda:0")
cfg = O.TINY._replace(num_layers=6, max_seq_len=2048)
p = O.init_params(cfg)
p.update(O.init_lora(cfg, 8))
pb, _ = bf16_params(p)
tokens = O.randint("tokens", (2, 2048), 0, cfg.vocab_size)
labels = torch.roll(tokens, -1, 1)
orig_save = ops._save
for ckpt in (False, True):
    refs = []

    def spy(ctx, *objs, _refs=refs):
        def walk(o):
            if isinstance(o, torch.Tensor):
                _refs.append((weakref.ref(o), tuple(o.shape), o.untyped_storage().nbytes()))
            elif isinstance(o, (tuple, list)):
                for x in o:
                    walk(x)
        walk(objs)
        return orig_save(ctx, *objs)

    ops._save = spy
    model = build_model(cfg._replace(activation_checkpointing=ckpt), pb, cuda, lora_rank=8)
    for n, q in model.named_parameters():
        q.requires_grad_("lora_" in n or n.endswith("_norm.weight") or n.startswith("tok_embeddings"))
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    loss = model(tokens.to(cuda), labels=labels.to(cuda))
    gc.collect()
    torch.cuda.synchronize()
    after_fwd = torch.cuda.memory_allocated() - base
    peak_fwd = torch.cuda.max_memory_allocated() - base
    alive = [(s, nb) for r, s, nb in refs if r() is not None]
    print(f"ckpt={ckpt}: after fwd {after_fwd / 2**20:.0f} MiB, peak fwd {peak_fwd / 2**20:.0f} MiB, saved tensors alive {len(alive)}/{len(refs)}, "
          f"alive bytes {sum(nb for _, nb in alive) / 2**20:.0f} MiB")
    if ckpt:
        from collections import Counter
        print("  alive shapes:", Counter(s for s, _ in alive).most_common(8))
    torch.cuda.reset_peak_memory_stats()
    loss.backward()
    torch.cuda.synchronize()
    print(f"          peak bwd {(torch.cuda.max_memory_allocated() - base) / 2**20:.0f} MiB, after bwd {(torch.cuda.memory_allocated() - base) / 2**20:.0f} MiB")
    gc.collect()
    seen, rows = set(), []
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                st = o.untyped_storage()
                if st.data_ptr() not in seen:
                    seen.add(st.data_ptr())
                    rows.append((st.nbytes(), tuple(o.shape), str(o.dtype), type(o).__name__))
        except Exception:  # noqa: BLE001
            pass
    rows.sort(reverse=True)
    from collections import Counter
    cnt = Counter()
    for nb, shp, dt, ty in rows:
        cnt[(shp, dt)] += nb
    print("   live CUDA storages after bwd: total %.0f MiB in %d storages" % (sum(r[0] for r in rows) / 2**20, len(rows)))
    for (shp, dt), nb in cnt.most_common(14):
        print(f"      {nb / 2**20:8.1f} MiB  {shp} {dt}")
    del model, loss
    gc.collect()
ops._save = orig_save
