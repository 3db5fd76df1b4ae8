"""N>1 path on CPU: world_size-2 gloo processes exercise the shard / broadcast / gather code
that the GPU run uses with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mtts import dist as mdist


def test_shard_indices_balanced_and_complete():
    lengths = [5, 100, 7, 50, 60, 1, 99, 3]
    for world in (1, 2, 3, 8):
        shards = mdist.shard_indices(lengths, world)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lengths)))
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1
    two = mdist.shard_indices(lengths, 2)
    loads = [sum(lengths[i] for i in s) for s in two]
    assert abs(loads[0] - loads[1]) <= 20
    assert mdist.shard_indices([], 4) == [[], [], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # weights: only rank 0 holds them
        sd = {"a.weight": torch.arange(12, dtype=torch.float32).reshape(3, 4),
              "b.weight": torch.ones(5, dtype=torch.bfloat16) * 3} if rank == 0 else {}
        got = dict(mdist.broadcast_state_dict(sd, "cpu"))
        assert got["a.weight"].shape == (3, 4) and float(got["a.weight"].sum()) == 66.0
        assert got["b.weight"].dtype == torch.bfloat16 and float(got["b.weight"].float().sum()) == 15.0
        # dialogues: item i "generates" a ramp of length 10*(i+1); item 3 fails
        items = [{"text": "x" * (3 * i + 1)} for i in range(7)]

        def run_local(its, idxs):
            return [None if i == 3 else torch.full((10 * (i + 1),), float(i)) for i in idxs]

        res = mdist.process_batch_sharded(items, run_local)
        if rank == 0:
            assert len(res) == 7 and res[3] is None
            for i, w in enumerate(res):
                if i != 3:
                    assert w.shape == (10 * (i + 1),) and float(w[0]) == float(i)
        else:
            assert res is None
        # metric reduction as bench.py does it
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == float(world)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_broadcast_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def _bench(*args, env=None):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(args), env=e, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts 2 ranks itself (here: --dry-run = the launcher,
    the rendezvous and the MAX/SUM reductions over gloo, no engine); rank 0 prints ONE line whose n_gpus is 2 and
    whose value is the sum over ranks divided by the slowest rank's time."""
    rc, out, err = _bench("--gpus", "2", "--dry-run", "--steps", "10", "--batch", "32")
    assert rc == 0, err[-2000:]
    assert len(out) == 1
    o = out[0]
    assert o["n_gpus"] == 2 and o["ranks"] == 2 and o["steps"] == 10
    assert abs(o["value"] - 2 * 32 * 10 * 8 / (10 * 4e-3 * 1.01)) < 1e-6 * o["value"]
    rc, out, err = _bench("--dry-run", "--steps", "10")                     # N=1: no launcher, no process group
    assert rc == 0 and out[0]["n_gpus"] == 1 and out[0]["ranks"] == 1


def test_bench_refuses_a_world_that_differs_from_gpus():
    rc, out, err = _bench("--gpus", "4", "--dry-run", env=dict(WORLD_SIZE="2", RANK="0", MASTER_ADDR="127.0.0.1",
                                                              MASTER_PORT=str(_free_port())))
    assert rc != 0 and not out and "--gpus 4" in err


# ---- the sharded inference entry point (inference_sharded.py: BASELINE configs[3]) ---------------------------------
class _Tok:
    pad_token_id = 151643

    def encode(self, s, add_special_tokens=True):
        return [min(ord(c), 151000) for c in s]


def _stub_process_batch(items, tokenizer, model, spt, device, system_prompt, start_idx, use_normalize=False, indices=None):
    """Stands in for generation_utils.process_batch: same return format; audio = a ramp whose length and value encode the
    item's job-wide index and the Philox row the model was told to use for it."""
    idx = [start_idx + i for i in range(len(items))] if indices is None else list(indices)
    rows = model.sample_rows if model.sample_rows is not None else list(range(len(items)))
    texts = [{"index": g, "original_text": it["text"], "normalized_text": None, "final_text": it["text"], "use_normalize": use_normalize}
             for g, it in zip(idx, items)]
    audio = [None if "fail" in it["text"] else
             {"audio_data": torch.full((1, 100 + 10 * g), float(r)), "sample_rate": spt.output_sample_rate, "index": g}
             for g, r, it in zip(idx, rows, items)]
    return texts, audio


def _sharded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        import inference_sharded as ish
        from mtts import synth, synth_codec
        from modeling_asteroid import AsteroidTTSInstruct, GenerationConfig
        from XY_Tokenizer.xy_tokenizer.model import XY_Tokenizer
        w, r, dev = ish.init_distributed(backend="gloo")
        assert (w, r) == (world, rank)
        loads = []

        def loader(model_path, spt_cfg, spt_ckpt, torch_dtype=None, attn_implementation=None):
            loads.append(rank)                                  # only rank 0 may read the checkpoints
            cfg = synth.tiny(hidden_size=128, intermediate_size=256, num_attention_heads=2, num_key_value_heads=1)
            sd = {k: torch.from_numpy(v).to(torch.bfloat16) for k, v in synth.synth_weights(cfg, 5).items()}
            model = AsteroidTTSInstruct.from_state_dict(cfg, sd, GenerationConfig(max_new_tokens=9, do_samples=[True] * 8,
                                                                                  layers=[{"top_k": 5}] * 8, eos_token_id=cfg["eos_token_id"]))
            import test_pipeline_gpu_helpers as h
            ccfg = synth_codec.reduced(dec_layers=1, voc_layers=1, adapter_layers=1)
            spt = XY_Tokenizer(h.generator_params(ccfg), {k: torch.from_numpy(v) for k, v in list(synth_codec.synth_weights(ccfg, 9).items())[:40]})
            return _Tok(), model.eval(), spt.eval()

        tok, model, spt = ish.load_model_sharded("m", "c", "k", device="cpu", loader=loader)
        assert loads == ([0] if rank == 0 else [])
        assert tok.encode("ab") == [97, 98] and spt.output_sample_rate == 24000 and spt.nq == 8
        assert model.config.hidden_size == 128 and model.generation_config.max_new_tokens == 9
        assert model.generation_config.layers[3] == {"top_k": 5} and model.dtype == "bf16"
        # the same bytes everywhere: checksums of both state dicts meet on rank 0
        sums = torch.tensor([sum(float(v.float().abs().sum()) for v in model._sd.values()), len(model._sd),
                             sum(float(v.float().abs().sum()) for v in spt._sd.values()), len(spt._sd)], dtype=torch.float64)
        all_s = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(all_s, sums)
        assert all(torch.equal(a, all_s[0]) for a in all_s) and sums[1] > 20 and sums[3] == 40
        items = [{"text": "x" * (5 + 7 * (i % 4))} for i in range(7)]
        items[4]["text"] = "fail" + items[4]["text"]
        texts, audio = ish.process_batch_sharded(items, tok, model, spt, "cpu", "sys", 100, run_local=_stub_process_batch)
        assert model.sample_rows is None                       # restored after the call
        if rank == 0:
            want_t, want_a = _stub_process_batch(items, tok, model, spt, "cpu", "sys", 100)
            assert texts == want_t
            assert len(audio) == 7 and audio[4] is None
            for i, (a, wa) in enumerate(zip(audio, want_a)):
                if i == 4:
                    continue
                assert a["index"] == 100 + i and a["sample_rate"] == 24000
                assert a["audio_data"].shape == wa["audio_data"].shape == (1, 100 + 10 * (100 + i))
                assert float(a["audio_data"][0, 0]) == float(i)         # drawn with its job-wide Philox row, whichever rank ran it
        else:
            assert texts is None and audio is None
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_rank_sharded_inference_entry_point():
    """inference_sharded.load_model_sharded + process_batch_sharded over 2 gloo ranks: rank 0 alone loads, both ranks end
    up with identical weights / configs / tokenizer, the batch is dealt by work, each rank's rows keep their job-wide
    Philox row ids, and rank 0 returns exactly what one process_batch over the whole batch returns."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_sharded_entry_point_without_a_process_group_is_process_batch():
    import inference_sharded as ish
    from types import SimpleNamespace
    model = SimpleNamespace(sample_rows=None)
    spt = SimpleNamespace(output_sample_rate=24000)
    items = [{"text": "abc"}, {"text": "fail"}, {"text": "defgh"}]
    got = ish.process_batch_sharded(items, _Tok(), model, spt, "cpu", "sys", 3, run_local=_stub_process_batch)
    want = _stub_process_batch(items, _Tok(), model, spt, "cpu", "sys", 3)
    assert got[0] == want[0] and got[1][1] is None and torch.equal(got[1][2]["audio_data"], want[1][2]["audio_data"])
    assert ish.estimate_work(items, _Tok(), "sys") == [3, 4, 5]
