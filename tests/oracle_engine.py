"""TEST-ONLY adapter: the CPU oracle behind the Engine method surface of dne_hip._lib, so that the host-side
logic (drivers, in-process transport, sharding, all-gather) can be exercised without a GPU.  Lives under
tests/ because only tests may call the oracle; the product never imports this."""
import numpy as np

import oracle as O


class OracleEngine:
    def __init__(self, kind, n_actions=18, max_members=64, ref_count=16, **kw):
        self.kind, self.n_actions, self.max_members, self.ref_count = kind, n_actions, max_members, ref_count
        self.L = O.layout(kind, n_actions)
        self.P = self.L.P
        self.bc_max_steps = kw.get("bc_max_steps", 0)
        self.bc_final_only = kw.get("bc_final_only", False)
        self.theta = np.zeros(self.P, np.float32)
        self.noise = None
        self.ref = None
        self.opt = None
        self.envs = [O.WrappedEnv() for _ in range(4)]
        self.members = None
        self.calls = []

    def close(self):
        pass

    # the parts of the Engine surface tools/workloads.py touches besides the evaluation calls
    def barrier(self):
        pass

    def check_redzones(self):
        return 0

    def profile(self):
        return {k: 0.0 for k in ("eval_ms", "fc_ms", "conv_ms", "env_ms", "ref_ms", "reduce_ms")}

    def comm_share(self, owner):
        self.shared_from = owner

    def noise_upload(self, noise):
        self.noise = np.ascontiguousarray(noise, np.float32)

    def set_theta(self, theta, slot=0):
        assert slot == 0
        self.theta = np.array(theta, np.float32)

    def get_theta(self, slot=0):
        return self.theta.copy()

    def set_ref_batch(self, ref):
        self.ref = np.ascontiguousarray(ref, np.uint8)

    def optimizer_reset(self):
        self.opt = None

    def set_members(self, slot, off, scale):
        self.members = (np.asarray(slot), np.asarray(off), np.asarray(scale, np.float32))

    def _member_theta(self, i):
        slot, off, scale = self.members
        return self.theta + np.float32(scale[i]) * self.noise[off[i]:off[i] + self.P]

    def es_eval(self, idx, sigma, tslimit, seeds, want_bc=False):
        self.calls.append(("es_eval", len(idx)))
        if self.bc_final_only and want_bc:
            n = len(idx)
            ret = np.zeros((n, 2), np.float32); sg = np.zeros((n, 2), np.float32); ln = np.zeros((n, 2), np.int32)
            bc = np.zeros((2 * n, 128), np.uint8)
            for i in range(n):
                for s in range(2):
                    th = O.perturb(self.theta, self.noise, idx[i], sigma, 1 if s == 0 else -1)
                    ret[i, s], sg[i, s], ln[i, s], traj = O.rollout(self.L, th, self.ref, seeds[2 * i + s], tslimit, want_bc=True)
                    bc[2 * i + s] = traj[-1]
            self._last = (np.asarray(idx, np.int64), ret, sg, ln)
            return ret, sg, ln, bc
        if not self.bc_max_steps:
            out = O.es_eval(self.L, self.theta, self.noise, idx, sigma, tslimit, self.ref, seeds)
            self._last = (np.asarray(idx, np.int64),) + tuple(out)
            return out
        n = len(idx)
        ret = np.zeros((n, 2), np.float32); sg = np.zeros((n, 2), np.float32); ln = np.zeros((n, 2), np.int32)
        self.bcs = []
        for i in range(n):
            for s in range(2):
                th = O.perturb(self.theta, self.noise, idx[i], sigma, 1 if s == 0 else -1)
                r, q, l, bc = O.rollout(self.L, th, self.ref, seeds[2 * i + s], tslimit, want_bc=True)
                ret[i, s], sg[i, s], ln[i, s] = r, q, l
                self.bcs.append(bc)
        return ret, sg, ln

    def eval_members(self, n, tslimit, seeds, want_bc=False):
        out = [O.rollout(self.L, self._member_theta(i), self.ref, seeds[i], tslimit, want_bc=want_bc) for i in range(n)]
        r = np.array([o[0] for o in out], np.float32); s = np.array([o[1] for o in out], np.float32)
        l = np.array([o[2] for o in out], np.int32)
        if not want_bc:
            return r, s, l
        if self.kind == O.KIND_ES and self.bc_final_only:
            return r, s, l, np.stack([o[3][-1] for o in out])
        if self.kind == O.KIND_ES:
            bc = np.zeros((n, max(self.bc_max_steps, int(l.max())), 128), np.uint8)
            for i, o in enumerate(out):
                bc[i, :l[i]] = o[3]
        else:
            bc = np.stack([o[3] for o in out])
        return r, s, l, bc

    def novelty(self, archive, bc, k):
        return O.novelty(archive, bc, k)

    def novelty_batch(self, archive, lengths, k):
        return np.array([O.novelty(archive, b, k) for b in self.bcs])

    def centered_ranks(self, x):
        x = np.asarray(x, np.float32)
        return O.centered_ranks(x.reshape(-1)).reshape(x.shape)

    def weighted_sum(self, idx, w, denom, copy_out=True):
        self.g = O.weighted_sum(self.noise, idx, w, self.P, denom)
        return self.g if copy_out else None

    def optimizer_step(self, kind, l2coeff, stepsize, beta1_or_momentum=0.9, beta2=0.999, epsilon=1e-8):
        if self.opt is None:
            self.opt = (O.Adam(self.theta, stepsize, beta1_or_momentum, beta2, epsilon) if kind == "adam"
                        else O.SGD(self.theta, stepsize, beta1_or_momentum))
        self.opt.theta = self.theta.copy()
        ratio, th = self.opt.update(self.g, l2coeff)
        self.theta = th.copy()
        return ratio

    def optimizer_get_state(self):
        if self.opt is None:
            return np.zeros(self.P, np.float32), np.zeros(self.P, np.float32), 0
        return getattr(self.opt, "m", np.zeros(self.P, np.float32)).copy(), self.opt.v.copy(), getattr(self.opt, "t", 0)

    def optimizer_set_state(self, m, v, t):
        self.opt = O.Adam(self.theta, 0.01)
        self.opt.m, self.opt.v, self.opt.t = np.array(m, np.float32), np.array(v, np.float32), int(t)
        self._adam_args_pending = True

    def ga_rebuild(self, slot, seeds, sigma, copy_out=True):
        th = O.ga_rebuild(self.L, self.noise, seeds, sigma)
        if slot == 0:
            self.theta = th.copy()
        return th

    def ga_eval(self, chains, sigma, tslimit, seeds, want_bc=False):
        self.calls.append(("ga_eval", len(chains)))
        out = [O.rollout(self.L, O.ga_rebuild(self.L, self.noise, c, sigma), None, seeds[i], tslimit)[:3]
               for i, c in enumerate(chains)]
        r, s, l = zip(*out)
        return np.array(r, np.float32), np.array(s, np.float32), np.array(l, np.int32)

    def ga_set_init_scale(self, scale_by):
        self.scale_by = np.asarray(scale_by, np.float32)

    def ga_rebuild_powers(self, slot, seeds, copy_out=True):
        return O.ga_gpu_rebuild(self.noise, seeds, self.scale_by)

    def ga_eval_powers(self, genomes, tslimit, seeds, want_bc=False):
        self.calls.append(("ga_eval_powers", len(genomes)))
        out = [O.rollout(self.L, O.ga_gpu_rebuild(self.noise, g, self.scale_by), None, seeds[i], tslimit)[:3] for i, g in enumerate(genomes)]
        r, s, l = zip(*out)
        return np.array(r, np.float32), np.array(s, np.float32), np.array(l, np.int32)

    def ga_select(self, returns, t):
        return O.ga_select(returns, t)

    def es_update(self, idx, returns_n2, signreturns_n2, proc_mode, opt_kind, l2coeff, stepsize,
                  beta1_or_momentum=0.9, beta2=0.999, epsilon=1e-8):
        self.calls.append(("es_update", len(idx)))
        rets = np.asarray(returns_n2, np.float32).reshape(-1, 2)
        if proc_mode == "centered_rank":
            proc = O.centered_ranks(rets.reshape(-1)).reshape(-1, 2)
        elif proc_mode == "sign":
            proc = np.asarray(signreturns_n2, np.float32).reshape(-1, 2)
        else:
            proc = O.centered_ranks(np.asarray(signreturns_n2, np.float32).reshape(-1)).reshape(-1, 2)
        g = O.weighted_sum(self.noise, idx, proc[:, 0] - proc[:, 1], self.P, float(rets.size))
        if self.opt is None:
            self.opt = (O.Adam(self.theta, stepsize, beta1_or_momentum, beta2, epsilon) if opt_kind == "adam"
                        else O.SGD(self.theta, stepsize, beta1_or_momentum))
        self.opt.theta = self.theta.copy()
        ratio, th = self.opt.update(g, l2coeff)
        self.theta = th.copy()
        return ratio

    # exchange surface (one rank: the all-gather is the identity)
    def records_pack(self, n_local):
        from dne_hip import _lib
        idx, ret, sg, ln = self._last
        assert len(idx) == n_local
        rec = np.zeros(n_local, _lib.RECORD)
        rec["noise_idx"], rec["ret"], rec["len"], rec["aux"] = idx, ret, ln, sg
        return rec

    def records_set(self, rec):
        self._rec = np.array(rec)

    def allgather_results(self, n_local, n_global):
        assert n_local == n_global
        self._rec = self.records_pack(n_local)
        return self._rec

    def es_update_gathered(self, proc_mode, opt_kind, l2coeff, stepsize, beta1_or_momentum=0.9, beta2=0.999, epsilon=1e-8):
        r = self._rec
        return self.es_update(r["noise_idx"], r["ret"], r["aux"], proc_mode, opt_kind, l2coeff, stepsize, beta1_or_momentum,
                              beta2, epsilon)

    # single-env ABI (slot 0) used by HipAtariEnv / Policy.rollout
    def env_reset(self, seeds):
        for e, s in zip(self.envs, seeds):
            e.reset(int(s))

    def env_step(self, actions):
        out = [self.envs[i].step(int(a)) for i, a in enumerate(np.atleast_1d(actions))]
        return np.array([o[1] for o in out], np.float32), np.array([o[2] for o in out], bool)

    def env_observation(self, n):
        return np.stack([self.envs[i].ob() for i in range(n)])

    def env_ram(self, n):
        return np.stack([self.envs[i].ram() for i in range(n)])

    def env_set_observation(self, obs):
        self._obs = np.asarray(obs, np.uint8)

    def ref_pass(self, n):
        if self.kind != O.KIND_ES:
            self._bn = [None] * n
            return
        both = [O.es_ref_pass_moments(self.L, self._member_theta(i), self.ref) for i in range(n)]
        self._bn, self._mom = [b for b, _ in both], [m for _, m in both]

    def get_bn_moments(self, n):
        return np.stack(self._mom[:n])

    def act(self, n):
        acts, lgs = [], []
        for i in range(n):
            ob = self._obs[i] if getattr(self, "_obs", None) is not None else self.envs[i].ob()
            a, lg = O.act(self.L, self._member_theta(i), self._bn[i] if self.kind == O.KIND_ES else None, ob)
            acts.append(a); lgs.append(lg)
        self._obs = None
        return np.array(acts, np.int32), np.stack(lgs)
