#!/usr/bin/env python3
"""Golden fixtures from the REAL reference for the widened rows (transport wire format, GPU-tree genomes and schedules).
Runs only in the build container (needs /root/reference); writes tests/golden/reference_wire.npz, which is committed.

Imported from the reference, unmodified (third-party modules absent here -- redis, tensorflow, the gym_tensorflow native op --
are stubbed ONLY so that the modules import; no value below comes from a stub):
  es_distributed/es.py, ga.py, es_modified.py      Config / Task / Result / GATask namedtuples, pickled as dist.py:19-20 does
  gpu_implementation/neuroevolution/helper.py       make_schedule, ConstantSchedule, LinearSchedule
  gpu_implementation/neuroevolution/models/base.py  BaseModel.compute_weights_from_seeds / compute_mutation / mutate / randomize
  gpu_implementation/neuroevolution/models/dqn.py   Model.create_weight_variable's scale_by formula
"""
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "gpu_implementation"))
for name in ("redis", "tensorflow", "tabular_logger", "gym_tensorflow", "gym_tensorflow.ops"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["gym_tensorflow.ops"].indexed_matmul = None
sys.modules["gym_tensorflow"].ops = sys.modules["gym_tensorflow.ops"]

import es_distributed.es as res              # noqa: E402
import es_distributed.ga as rga              # noqa: E402
import es_distributed.es_modified as rmod    # noqa: E402
import es_distributed.dist as rdist          # noqa: E402
from neuroevolution import helper            # noqa: E402
from neuroevolution.models import base, dqn  # noqa: E402


def main():
    out = {}
    # ---- what the reference puts on the wire (dist.py:19-20 serialize = pickle protocol -1)
    task = res.Task(params=np.arange(5, dtype=np.float32), ob_mean=None, ob_std=None, ref_batch=[np.zeros((84, 84, 4), np.float32)], timestep_limit=7)
    gtask = rga.GATask(params=np.ones(3, np.float32), population=[[1, 2], [3]], ob_mean=None, ob_std=None, timestep_limit=9)
    result = res.Result(worker_id=3, noise_inds_n=np.array([11, 12]), returns_n2=np.ones((2, 2), np.float32),
                        signreturns_n2=np.ones((2, 2), np.float32), lengths_n2=np.ones((2, 2), np.int32), eval_return=None,
                        eval_length=None, ob_sum=None, ob_sumsq=None, ob_count=0)
    mres = rmod.Result(worker_id=4, noise_inds_n=np.array([5]), returns_n2=np.zeros((1, 2), np.float32), signreturns_n2=np.zeros((1, 2), np.float32),
                       lengths_n2=np.ones((1, 2), np.int32), eval_return=None, eval_length=None, ob_sum=None, ob_sumsq=None, ob_count=0,
                       bc_vectors=[(np.arange(128, dtype=np.uint8)[None], 10.0, 3, 5, 777, 1)])
    cfg = res.Config(l2coeff=0.005, noise_stdev=0.02, episodes_per_batch=10, timesteps_per_batch=10, calc_obstat_prob=0.0, eval_prob=0.0,
                     snapshot_freq=0, return_proc_mode="centered_rank", episode_cutoff_mode=5)
    for k, v in (("task", task), ("gatask", gtask), ("result", (4, result)), ("modified_result", (6, mres)), ("config", cfg)):
        out["wire_" + k] = np.frombuffer(rdist.serialize(v), np.uint8)
    out["wire_keys"] = np.array([rdist.EXP_KEY, rdist.TASK_ID_KEY, rdist.TASK_DATA_KEY, rdist.TASK_CHANNEL, rdist.RESULTS_KEY, rdist.ARCHIVE_KEY])
    # ---- schedules (helper.py:46-88; ExponentialSchedule.value raises in the reference -- self.linear is not callable -- so it
    #      has no golden values)
    its = np.array([0, 1, 5, 10, 50])
    lin = helper.make_schedule({"type": "LinearSchedule", "schedule": 10, "initial_p": 0.01, "final_p": 0.001, "field": "iteration"})
    out["sched_iterations"] = its
    out["sched_linear"] = np.array([lin.value(iteration=int(i), timesteps_so_far=0) for i in its])
    out["sched_constant"] = np.array([helper.make_schedule(0.002).value(iteration=3)])
    # ---- GPU-tree genomes: the real compute_weights_from_seeds on a bare BaseModel carrying Model's scale_by
    nact, shapes = 18, [(8, 8, 4, 16), (1, 1, 1, 16), (4, 4, 16, 32), (1, 1, 1, 32), (3872, 256), (256,), (256, 18), (18,)]
    m = dqn.Model.__new__(dqn.Model)
    captured = []
    m.create_variable = lambda name, shape, scale_by: captured.append(scale_by)     # dqn.py:24-27 computes scale_by, then calls this
    scale = []
    for i, shp in enumerate(shapes):
        if i % 2 == 0:
            dqn.Model.create_weight_variable(m, "w", shp, 0.1 if i == 6 else 1.0)
            scale.append(np.full(int(np.prod(shp)), captured[-1], np.float32))
        else:
            scale.append(np.zeros(int(np.prod(shp)), np.float32))                    # base.py:49-50 create_bias_variable: scale_by 0.0
    scale_by = np.concatenate(scale)
    m.scale_by, m.num_params = scale_by, scale_by.size

    class Noise:   # helper.py:27-43 SharedNoiseTable get / sample_index over a small table of the reference stream
        noise = np.random.RandomState(123).randn(4_000_000).astype(np.float32)

        def get(self, i, dim):
            return self.noise[i:i + dim]

        def sample_index(self, stream, dim):
            return stream.randint(0, len(self.noise) - dim + 1)
    noise = Noise()
    rs = np.random.RandomState(11)
    theta0, seeds0 = m.randomize(rs, noise)
    theta1, seeds1 = m.mutate((theta0, seeds0), rs, noise, mutation_power=0.002)
    theta2, seeds2 = m.mutate((theta1, seeds1), rs, noise, mutation_power=0.004)
    assert np.array_equal(m.compute_weights_from_seeds(noise, seeds2), theta2)
    sel = np.r_[0:64, 4096:4120, 12400:12464, 500_000:500_064, scale_by.size - 80:scale_by.size]
    out["gpu_scale_by_sel"] = scale_by[sel]
    out["gpu_sel"] = sel
    out["gpu_seeds_idx"] = np.array([seeds2[0], seeds2[1][0], seeds2[2][0]], np.int64)
    out["gpu_seeds_power"] = np.array([0.0, seeds2[1][1], seeds2[2][1]])
    for i, th in enumerate((theta0, theta1, theta2)):
        out["gpu_theta%d_sel" % i] = np.asarray(th, np.float32)[sel]
        out["gpu_theta%d_dtype" % i] = np.array(str(np.asarray(th).dtype))
        out["gpu_theta%d_sum" % i] = np.array(np.asarray(th, np.float64).sum())
    path = os.path.join(HERE, "reference_wire.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "keys:", len(out), "bytes:", os.path.getsize(path))


if __name__ == "__main__":
    main()
