"""BASELINE config 5: the homomorphic PRINCE block (arrays client; the reference client's gate-by-gate pattern, synchronous / scheduled / library default)."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

from .placement import restore_original_affinity
from .record import HBM_PEAK_GBS, ROOT


def bench_prince(world, single_dev):
    """BASELINE config 5: wall clock of one homomorphic PRINCE block (examples/Prince/Prince.cu:83-87 times princeEncrypt)
    and its known answer (Prince.cu:96), gates on arrays of ciphertexts, S-boxes of a layer spread over `world` GPUs by
    the in-process multi-device driver (tests/cxx/test_prince_arrays_cxx.cpp)."""
    try:
        exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_prince_arrays_cxx")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cuhe_amd", "cxx"), "-s", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        # three blocks in one process: `value` is the FIRST (what the reference times: one block after set-up, examples/Prince/Prince.cu:83-87);
        # the later ones no longer pay the first-time hipMalloc of the arrays (profiles/r05_prince_gaps_arrays.txt)
        # (one GPU only: the repeated form of the multi-device client has not run on hardware)
        cmd = [exe, "--no-round-checks", "--async", "--json"] + (["--repeat", "3"] if world == 1 else []) + ["--devices", str(world)] + (["--virtual"] if single_dev and world > 1 else [])
        # (N > 1: the client's threads drive GPUs on both sockets and place themselves per device -- not inside the mask of rank 0's own GPU)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, preexec_fn=restore_original_affinity if world > 1 else None)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stdout[-300:] + r.stderr[-300:]).strip()}
        recs = [json.loads(l) for l in line]
        rec = recs[0]
        out = {"value": rec["prince_seconds"], "unit": "s per PRINCE block (64 ciphertext bits, 1920 cAnd, 1152 relin, 24 levels)", "n_gpus": rec["devices"],
               "virtual_devices": rec["virtual"], "known_answer": rec["kat"], "known_answer_ok": rec["kat_ok"],
               "params": "CuDHS(25,2,16,25,25,21845): n=16384, 32K-point transforms, 25 -> 1 primes, 40 keys",
               "client": "CuCtxtArray (not the reference's call pattern)",
               "mode": "CuCtxtArray gates, asynchronous, one host thread per GPU (Prince.cu:194-200)",
               "later_blocks_same_process": [x["prince_seconds"] for x in recs[1:]],
               "all_known_answers_ok": all(x["kat_ok"] for x in recs)}
        out["gate_by_gate"] = bench_prince_gate_by_gate(literal=world == 1)     # (the literal-client leg: 35 s of one-GPU work, at N = 1 only)
        return out
    except Exception as ex:
        return {"error": repr(ex)[:300]}


def bench_prince_gate_by_gate(literal=True):
    """The reference client's call pattern (examples/Prince/Prince.cu:204-322: ONE host thread, the default stream, one
    CuCtxt gate per call) on one GPU, same block and known answer: with the reference's synchronise-per-gate semantics
    (cuhe/CuHE.cu:98,121,139,157) and with the library's scheduled gates (CuHE.h setScheduled / CUHE_SCHED=1: the same
    client code, independent gates issued concurrently by the library's worker threads)."""
    try:
        exe = os.path.join(ROOT, "cuhe_amd", "lib", "test_prince_flow")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "cuhe_amd", "cxx"), "-s", exe], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        # one synchronous block, then four scheduled blocks in the same process: the first is the figure of rounds 4 (workers and their
        # scratch are new), the later ones are what a client that encrypts more than one block sees
        r = subprocess.run([exe, "--threads", "1", "--no-round-checks", "--compare", "--repeat", "4"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, CUHE_SCHED_STATS="1"))
        secs, blocks = {}, []
        for l in r.stdout.splitlines():
            if l.startswith("Prince Encryption:"):
                if "scheduled gates" in l:
                    blocks.append(float(l.split()[2]))
                else:
                    secs["sync_1thread"] = float(l.split()[2])
            elif l.startswith("batches:"):
                secs["scheduler"] = l.strip()[:200]
        if blocks:
            # `scheduled_1thread` is the FIRST scheduled block of the process: what the reference times (one block after set-up,
            # examples/Prince/Prince.cu:83-87) and what the synchronous figure and rounds 1-4 are.  It pays the workers'
            # first-time scratch (several GB of hipMalloc); the median of the blocks after it is printed under its own name.
            later = sorted(blocks[1:])
            secs["scheduled_1thread"] = blocks[0]
            secs["scheduled_1thread_warm_median"] = later[len(later) // 2] if later else None
            secs["scheduled_1thread_blocks"] = blocks
        ok = r.returncode == 0 and r.stdout.count("9fb51935fc3df524   expected 9fb51935fc3df524   right") == 5 and len(blocks) == 4 and "sync_1thread" in secs
        if not ok:
            return {"error": (r.stdout[-300:] + r.stderr[-300:]).strip()}
        secs.update({"unit": "s per PRINCE block, CuCtxt gates one per call from one host thread (the reference client's pattern)", "known_answer_ok": True})
        # the UNCHANGED client: no setScheduled call, no CUHE_SCHED in the environment -- what initCuHE gives by default since round 6
        # (scheduled gates); two blocks in one process
        try:
            env = dict(os.environ); env.pop("CUHE_SCHED", None)
            r2 = subprocess.run([exe, "--threads", "1", "--no-round-checks", "--default", "--repeat", "2"], capture_output=True, text=True, timeout=900, env=env)
            lines = [l for l in r2.stdout.splitlines() if l.startswith("Prince Encryption:")]
            ok2 = r2.returncode == 0 and len(lines) == 2 and r2.stdout.count("9fb51935fc3df524   expected 9fb51935fc3df524   right") == 2
            secs["library_default_1thread"] = ({"first_block": float(lines[0].split()[2]), "second_block": float(lines[1].split()[2]),
                                                "gates": "scheduled" if "scheduled gates" in lines[0] else "synchronous", "environment": "CUHE_SCHED unset, no setScheduled call"}
                                               if ok2 else {"error": (r2.stdout[-200:] + r2.stderr[-200:]).strip()})
        except Exception as ex:
            secs["library_default_1thread"] = {"error": repr(ex)[:200]}
        # the reference example's LITERAL structure (Prince.cu:188-322): the state lives on the host as ZZX between S-boxes, every S-box hands four
        # ZZX in and takes four back, 8 client threads (one per S-box in flight).  Library default against CUHE_SCHED=0; the host linear layers (fallback
        # big integer) are excluded from the time by the program, the client's own ZZX copies are not.
        try:
            if not literal:
                return secs
            lit = {"unit": "s per PRINCE block, ZZX state on the host between S-boxes (test_prince_flow --zzx-state), 8 client threads", "blocks_per_process": 3}
            for name, extra in (("library_default", {}), ("synchronous", {"CUHE_SCHED": "0"})):
                env = dict(os.environ); env.pop("CUHE_SCHED", None); env.update(extra)
                r3 = subprocess.run([exe, "--threads", "8", "--zzx-state", "--no-round-checks", "--default", "--repeat", "3"], capture_output=True, text=True, timeout=900, env=env)
                lines = [l for l in r3.stdout.splitlines() if l.startswith("Prince Encryption:")]
                ok3 = r3.returncode == 0 and len(lines) == 3 and r3.stdout.count("9fb51935fc3df524   expected 9fb51935fc3df524   right") == 3
                lit[name] = sorted(float(l.split()[2]) for l in lines)[1] if ok3 else {"error": (r3.stdout[-200:] + r3.stderr[-200:]).strip()}
            secs["literal_client_zzx_state_8threads"] = lit
        except Exception as ex:
            secs["literal_client_zzx_state_8threads"] = {"error": repr(ex)[:200]}
        return secs
    except Exception as ex:
        return {"error": repr(ex)[:300]}
