"""CPU: bench.py's own N > 1 launcher (the launcher side of gpu_implementation/neuroevolution/concurrent_worker.py:129-142) and
the two rendezvous forms of its ranks, exercised with stand-in ranks that speak the protocol but own no GPU: the RCCL id reaches
every rank, the carrier is agreed by ALL ranks (one failed or silent rank moves everybody to gloo), a dying rank ends the launch."""
import argparse
import json
import os
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
rdv = bench.PipeRendezvous(os.environ["DNE_CTRL_FDS"])
mode = os.environ.get("STUB_MODE", "ok")
uid, err = rdv.exchange_uid(rank, (bytes(range(128)) if mode != "no_uid" else None) if rank == 0 else None, "cannot open librccl" if mode == "no_uid" else None)
if mode == "die" and rank == 1:
    sys.exit(7)
ok = uid == bytes(range(128)) and not (mode == "fail1" and rank == 1)
d = rdv.vote(rank, world, ok, None if ok else (err or "ncclCommInitRank: unhandled system error"))
json.dump({"uid_ok": uid == bytes(range(128)), "decision": d}, open(os.path.join(os.environ["STUB_OUT"], "r%%d.json" %% rank), "w"))
if rank == 0:
    print(json.dumps({"metric": "stub", "carrier": d["carrier"]}))
"""


def _launch(tmp_path, mode, n=3):
    sys.path.insert(0, ROOT)
    import bench
    os.environ["STUB_MODE"], os.environ["STUB_OUT"] = mode, str(tmp_path)
    try:
        args = argparse.Namespace(gpus=n, single_device=False, transport="rccl")
        rc = bench.launch_ranks(args, child_argv=[sys.executable, "-c", STUB % {"root": ROOT}], check_devices=False)
    finally:
        del os.environ["STUB_MODE"], os.environ["STUB_OUT"]
    out = {}
    for r in range(n):
        p = tmp_path / ("r%d.json" % r)
        if p.exists():
            out[r] = json.load(open(p))
    return rc, out


@pytest.mark.timeout(120)
def test_self_launcher_carries_the_id_and_the_unanimous_vote(tmp_path, capfd):
    rc, out = _launch(tmp_path, "ok")
    assert rc == 0 and sorted(out) == [0, 1, 2]
    assert all(o["uid_ok"] and o["decision"] == {"carrier": "rccl", "errors": []} for o in out.values())
    assert '"carrier": "rccl"' in capfd.readouterr().out          # rank 0's stdout is the launcher's


@pytest.mark.timeout(120)
def test_one_failed_rank_moves_every_rank_to_gloo(tmp_path):
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    rc, out = _launch(tmp_path / "a", "fail1")
    assert rc == 0 and sorted(out) == [0, 1, 2]
    for o in out.values():
        assert o["decision"]["carrier"] == "gloo" and o["decision"]["errors"] == ["rank 1: ncclCommInitRank: unhandled system error"]
    rc, out = _launch(tmp_path / "b", "no_uid")
    assert rc == 0 and all(o["decision"]["carrier"] == "gloo" and not o["uid_ok"] for o in out.values())
    assert all("cannot open librccl" in e for e in out[0]["decision"]["errors"])


@pytest.mark.timeout(120)
def test_a_dying_rank_ends_the_launch_with_its_code(tmp_path):
    rc, out = _launch(tmp_path, "die")
    assert rc == 7
    assert all(o["decision"]["carrier"] == "gloo" for o in out.values())      # the survivors were told, none hung in RCCL


def test_launcher_refuses_when_devices_are_missing(capfd):
    sys.path.insert(0, ROOT)
    import bench
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    assert bench.launch_ranks(argparse.Namespace(gpus=2, single_device=False, transport="rccl")) == 2
    err = capfd.readouterr().err
    assert "needs 2 HIP devices" in err and "--single-device --transport gloo" in err


@pytest.mark.timeout(120)
def test_file_rendezvous_of_an_external_launcher(tmp_path, monkeypatch):
    """torch.distributed.run's ranks: id and votes through files in a private directory; stale files are ignored"""
    sys.path.insert(0, ROOT)
    import bench
    world, res = 3, {}

    def rank_fn(r, fail):
        rdv = bench.FileRendezvous(world)
        uid, err = rdv.exchange_uid(r, bytes(range(128)) if r == 0 else None, None)
        res[r] = (uid, rdv.vote(r, world, not (fail and r == 2), "boom" if fail and r == 2 else None))
    for fail in (False, True):
        # every launch has a directory of its own (keyed on the launcher's pid, start time and restart count): a second "launch" into the
        # first one's directory would let a rank read the FIRST launch's still-fresh id before rank 0 has cleared it -- a race this test
        # used to lose now and then, and one a real launch cannot have
        d = tmp_path / ("rdv%d" % int(fail))
        os.makedirs(d)
        monkeypatch.setenv("DNE_RDV_DIR", str(d))
        stale = d / "uid"
        stale.write_text(json.dumps({"uid": "00" * 128, "err": None}))
        os.utime(stale, (1, 1))                                          # a leftover from long ago
        res.clear()
        ts = [threading.Thread(target=rank_fn, args=(r, fail)) for r in range(world)]
        [t.start() for t in ts]
        [t.join(60) for t in ts]
        assert all(res[r][0] == bytes(range(128)) for r in range(world))
        want = {"carrier": "gloo", "errors": ["rank 2: boom"]} if fail else {"carrier": "rccl", "errors": []}
        assert all(res[r][1] == want for r in range(world))


def test_rccl_is_not_optional_by_default():
    """--transport rccl: a vote that ends on gloo fails the launch (exit 3) unless --allow-gloo-fallback was given"""
    sys.path.insert(0, ROOT)
    import bench
    lost = {"carrier": "gloo", "errors": ["rank 1: ncclCommInitRank: unhandled system error"]}
    msg = bench.gloo_fallback_refusal(lost, allowed=False)
    assert msg and "rank 1: ncclCommInitRank" in msg and "--allow-gloo-fallback" in msg and bench.EXIT_NO_RCCL == 3
    assert bench.gloo_fallback_refusal(lost, allowed=True) is None
    assert bench.gloo_fallback_refusal({"carrier": "rccl", "errors": []}, allowed=False) is None


def test_file_rendezvous_ignores_votes_of_another_launch(tmp_path, monkeypatch):
    """a vote file without this launch's token (rank 0's nonce, published with the id) is not read as a vote"""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("DNE_RDV_DIR", str(tmp_path / "rdv"))
    os.makedirs(tmp_path / "rdv")
    r0, r1 = bench.FileRendezvous(2), bench.FileRendezvous(2)
    (tmp_path / "rdv" / "vote.1").write_text(json.dumps({"ok": True, "err": None, "token": "feedfacefeedface"}))   # fresh, but foreign
    uid0, _ = r0.exchange_uid(0, bytes(range(128)), None)        # rank 0 clears leftovers, then publishes id + token
    uid1, _ = r1.exchange_uid(1, None, None)
    assert uid0 == uid1 == bytes(range(128)) and r0.token == r1.token and len(r0.token) == 16
    (tmp_path / "rdv" / "vote.1").write_text(json.dumps({"ok": True, "err": None, "token": "feedfacefeedface"}))
    assert r0._get("vote.1", 0.2, "rank 1's vote", token=r0.token) is None
    res = {}
    ts = [threading.Thread(target=lambda r=r, v=v: res.update({r: v.vote(r, 2, True, None)})) for r, v in ((0, r0), (1, r1))]
    [t.start() for t in ts]
    [t.join(30) for t in ts]
    assert res[0] == res[1] == {"carrier": "rccl", "errors": []}
