"""Rollout collection over the time-sliced engine (include/grip_sim.h: grip_batch_advance).

One env macro step (RobotEnv.step, robot_env.py:77-241) costs 17..1200 calls of physics.step() depending on the
action, so a lock-step vector env spends most of a launch waiting for its slowest member. Here every env runs on its
own clock: each tick gives all envs a bounded slice of physics, the envs that finished are listed (at most `capacity`
per tick), only those are rendered and fed to the policy, and their actions go back in with the next tick.

Transitions therefore arrive out of order. They are stored as *decision records* in HBM:

    record r = (env, obs, action, log_prob, value)         written when the policy decides for a listed env
    reward[r], done[r], next_rec[r]                        filled when that env is listed the next time

Record ids grow with time (N carry rows, then `capacity` rows per tick, in one of two alternating windows), so next_rec[r] > r and GAE is a
backward sweep over tick blocks with a gather through next_rec. A record that is still in flight when the rollout is
full bootstraps its predecessor with its value (as PPO's last_values do) and is carried into the carry rows of the next
rollout -- one record per env at most. A tick is fixed-shape, fixed-address device work with no host sync, and on a GPU
it is captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed: ~100 small launches become one.

The engine is duck-typed (`advance`, `observe_list`, `num_envs`, `action_dim`, `obs_shape`, `device`) so that the
bookkeeping is testable on CPU against a scripted engine (tests/test_async_rollout_cpu.py).
"""
import torch as th


class BatchEngineAdapter:
    """engine.Batch (via BatchedRobotEnv / GpuVecEnv) as the duck-typed async engine."""

    def __init__(self, env, budget_us=0):
        self.budget_us = int(budget_us)                  # wall-clock cap of a wavefront's slice (grip_sim.h), 0 = none
        benv = getattr(env, "env", env)                  # GpuVecEnv -> BatchedRobotEnv
        self.batch = benv.batch
        if not self.batch.cfg.auto_reset:
            self.batch.set_config(auto_reset=1)
        self.num_envs, self.action_dim, self.device = self.batch.n, self.batch.action_dim, self.batch.device
        self.obs_shape = (self.batch.obs_channels, 64, 64)
        self.im_reward = bool(getattr(benv, "im_reward", False))

    def add_intrinsic_reward(self, obs_records, old_rows, obs_rows, ready_list, ready_count, reward):
        """--im_reward: reward of every listed env += intrinsic_reward(obs of its previous decision record, new obs)."""
        self.batch.add_intrinsic_reward(obs_records, obs_rows, reward, old_rows=old_rows, ready_list=ready_list, ready_count=ready_count)

    def reset(self):
        self.batch.reset()

    def advance(self, slot_actions, slice_len, ready_list, ready_count, lag=1):
        return self.batch.advance(slot_actions, slice_len, ready_list, ready_count, self.budget_us, lag)

    def observe_list(self, ready_list, ready_count, obs_rows, records=None, record_row=None):
        self.batch.observe_list(ready_list, ready_count, obs_rows, records, record_row)


class AsyncRollout:
    def __init__(self, engine, policy_fn, target, capacity, slice_len, gamma, gae_lambda, max_ticks=None, action_low=None, action_high=None,
                 poll_every=8, use_graph=True, fused=None, pipeline=None, policy_parts_fn=None):
        """policy_fn(obs_rows uint8 [C, ...]) -> (actions [C, A], values [C], log_probs [C]) under no_grad.
        target = completed transitions per rollout; capacity = ready-list rows per tick."""
        self.eng, self.policy_fn = engine, policy_fn
        self.policy_parts_fn = policy_parts_fn          # obs rows -> (mean, log_std, values): sampling fused into the recorder kernel
        self.N, self.C, self.S, self.A = engine.num_envs, int(capacity), int(slice_len), engine.action_dim
        self.target, self.gamma, self.lam = int(target), gamma, gae_lambda
        dev = self.dev = engine.device
        if max_ticks is None:                            # generous: 4x the ticks a full ready list would need, at least 64
            max_ticks = max(64, 4 * (self.target + self.C - 1) // self.C)
        self.max_ticks = int(max_ticks)
        # two record windows (carry rows + max_ticks tick blocks each), used by alternate rollouts, so that an update may
        # still read rollout i while rollout i + 1 is being recorded; row R is a dump row for masked scatters
        self.W = self.N + self.max_ticks * self.C
        self.R = 2 * self.W
        self.win = 1                                      # the first _begin() flips to window 0
        R1 = self.R + 1
        self.obs = th.zeros((R1,) + tuple(engine.obs_shape), dtype=th.uint8, device=dev)
        self.actions = th.zeros(R1, self.A, device=dev)
        z = lambda dt=th.float32: th.zeros(R1, dtype=dt, device=dev)
        self.log_probs, self.values, self.rewards, self.dones, self.advantages, self.returns = z(), z(), z(), z(), z(), z()
        self.next_rec = th.full((R1,), -1, dtype=th.int64, device=dev)
        self.prev_rec = th.full((R1,), -1, dtype=th.int64, device=dev)
        self.is_rec = z(th.bool); self.completed = z(th.bool)
        self.rec_env = th.full((R1,), -1, dtype=th.int64, device=dev)
        self.rec_of_env = th.full((self.N + 1,), -1, dtype=th.int64, device=dev)      # row N: dump
        self.lst = th.full((self.C,), -1, dtype=th.int32, device=dev)
        self.cnt = th.zeros(1, dtype=th.int32, device=dev)
        self.slot_act = th.zeros(self.C, self.A, device=dev)
        self.ar_c = th.arange(self.C, device=dev)
        self.n_completed = th.zeros(1, dtype=th.int64, device=dev)
        self.low = None if action_low is None else th.as_tensor(action_low, device=dev, dtype=th.float32)
        self.high = None if action_high is None else th.as_tensor(action_high, device=dev, dtype=th.float32)
        self.poll_every = poll_every
        self.tick = 0                                    # ticks of the current rollout
        self.total_ticks = 0
        # episode statistics (device side; read with stats())
        self.ep_ret = th.zeros(self.N + 1, device=dev); self.ep_len = th.zeros(self.N + 1, device=dev)
        self.ep_ret_sum = th.zeros(1, device=dev); self.ep_len_sum = th.zeros(1, device=dev); self.ep_count = th.zeros(1, device=dev)
        self.substeps_total = th.zeros(1, dtype=th.int64, device=dev)          # physics.step() calls of the finished macro steps
        self.base_t = th.full((1,), self.N, dtype=th.int64, device=dev)        # first record row of the current tick (set by _begin)
        self.obs_stage = th.zeros((self.C,) + tuple(engine.obs_shape), dtype=th.uint8, device=dev)
        import os
        if os.environ.get("GRIP_ASYNC_GRAPH") == "0":
            use_graph = False
        self.use_graph, self.graph_after, self.eager_every, self._graph = use_graph, 3, 16, None
        # bookkeeping: one fused HIP launch per tick on a GPU (csrc/grip_rollout.hip), tensor ops otherwise (CPU tests; the
        # two are compared on the GPU by tests/test_gpu_async.py)
        self.fused = (dev.type == "cuda") if fused is None else bool(fused)
        self._targs = [None, None]
        # pipeline: decisions for the envs listed by tick t are made on a side stream WHILE tick t+1 advances everybody else
        # (grip_batch_advance lag = 2); two alternating sets of list / action / staging buffers, ordered with events
        # Measured on MI355X (tools/overlap_test2.py): the physics kernel already keeps every SIMD's VALU busy, the side work
        # only fills its latency gaps, and the extra tick of lag costs more than that gains -- off by default.
        self.pipeline = False
        self._started = False
        if pipeline:
            self.enable_pipeline()
        if self.low is None:
            self.low = th.full((self.A,), -float("inf"), device=dev); self.high = th.full((self.A,), float("inf"), device=dev)
        # slice schedule that follows the workload (set_slice_ladder): (mean physics.step() calls per macro step above which the
        # rung applies, slice, budget_us) -- long macro steps want long slices (fewer ticks, each paying the serial decision phase)
        self.ladder = None
        self._sub_seen = 0
        self.mean_substeps = None
        # synthetic action stream (set_synthetic_actions): per-env decision counters t of the counter-based generator
        self.rng_seed = None
        self.rng_count = th.zeros(self.N + 1, dtype=th.int64, device=dev)

    def set_synthetic_actions(self, seed, rank=0):
        """actions a ~ U(-1, 1)^A from a counter-based generator keyed by (seed, rank, env, t) instead of samples of
        the policy -- the benchmark's synthetic input: the physics workload then does not drift as the policy trains. The policy
        still runs on every listed env and the records hold the log-probability of the given action under it, so the PPO update does
        the same arithmetic. GPU fused recorder only. seed=None switches back to policy samples."""
        if seed is not None and not (self.fused and self.policy_parts_fn is not None):
            raise RuntimeError("synthetic actions need the fused recorder (GPU) and a policy with forward_parts")
        self.rng_seed = None if seed is None else ((int(seed) * 1000003 + int(rank)) * 0x9E3779B97F4A7C15 + 0x1234567) % (1 << 64)
        self.rng_count.zero_()
        self._graph = None
        self._targs = [None, None]

    def set_target(self, target):
        """Completed transitions per rollout; the tick budget follows (it was sized for the constructor's target)."""
        self.target = int(target)
        need = max(64, 4 * (self.target + self.C - 1) // self.C)
        if need > self.max_ticks:
            raise ValueError(f"target {target} needs {need} ticks of record rows, the rollout was built for {self.max_ticks}")

    def reset_counters(self):
        """Zero the statistics a benchmark reads over a timed region (keeps the slice ladder's bookkeeping consistent with them)."""
        self.substeps_total.zero_(); self._sub_seen = 0

    def set_slice_ladder(self, ladder=((0, 96, 2000), (215, 120, 2500), (270, 192, 4000))):
        """Let the slice length / wall-clock budget follow the measured mean number of physics.step() calls per macro step: after
        every rollout the rung with the largest threshold below that mean is taken (10 % hysteresis); a change re-captures the
        tick graph. Measured on the bench workload: 96 / 2000 us is best at ~180 calls per macro step (fresh episodes), 144-192 /
        3000-4000 us at ~300 (steady state of training): +9 % there, -16 % if used on the former. (Round 4: the middle rung 144 / 3000 -> 120 / 2500 -- the
        decision phase a tick pays for got 40 % shorter, and at ~236 calls per macro step 120 / 2500 and 96 / 2000 now measure +0.7 ... +1 % over 144 / 3000.)"""
        self.ladder = tuple(sorted((int(a), int(b), int(c)) for a, b, c in ladder))

    def _retune(self, done_n):
        if self.ladder is None or self.pipeline or done_n <= 0 or not hasattr(self.eng, "budget_us"):
            return
        tot = int(self.substeps_total.item())
        self.mean_substeps = (tot - self._sub_seen) / done_n
        self._sub_seen = tot
        cur = next((i for i, r in enumerate(self.ladder) if r[1] == self.S and r[2] == self.eng.budget_us), None)
        want = max(i for i, r in enumerate(self.ladder) if self.mean_substeps >= r[0] or i == 0)
        if cur is not None and want != cur:                   # hysteresis: move only when clearly past the threshold
            edge = self.ladder[max(want, cur)][0]
            if abs(self.mean_substeps - edge) < 0.1 * edge:
                want = cur
        if cur is None or want != cur:
            _, self.S, self.eng.budget_us = self.ladder[want]
            self._graph = None                                # slice and budget are launch arguments: capture the tick again

    def enable_pipeline(self):
        dev = self.dev
        if not self.pipeline:
            self.pipeline = True
            self.lst2 = [self.lst, th.full_like(self.lst, -1)]; self.cnt2 = [self.cnt, th.zeros_like(self.cnt)]
            self.slot_act2 = [self.slot_act, th.zeros_like(self.slot_act)]; self.obs_stage2 = [self.obs_stage, th.zeros_like(self.obs_stage)]
            self.side = th.cuda.Stream(dev); self.ev_side = [None, None]; self._side_graph = [None, None]

    # ------------------------------------------------------------------ one tick
    def _tick_body(self):
        """All of a tick as fixed-shape, fixed-address device work (the record rows of the tick come from the device scalar
        base_t), so that on a GPU the whole tick is one hipGraph launch."""
        out = self.eng.advance(self.slot_act, self.S, self.lst, self.cnt)
        self._decide(out, self.lst, self.cnt, self.obs_stage, self.slot_act, 0)

    def _decide(self, out, lst, cnt, obs_stage, slot_act, p):
        """Render the listed envs, run the policy on them, record the decisions; writes slot_act for the next start."""
        dual = getattr(self.eng, "batch", None) is not None       # the GPU engine writes the record rows itself
        im = getattr(self.eng, "im_reward", False)
        # render ONCE, straight into the tick's record rows, when the policy's first layer can read them there (engine.RecordRows): no
        # staging copy of the observations is written or read (the intrinsic reward still wants the staged rows)
        rows_only = bool(dual and not im and self.fused and self.policy_parts_fn is not None and getattr(self.policy_parts_fn, "accepts_record_rows", lambda: False)())
        if rows_only:
            from ..engine import RecordRows
            self.eng.observe_list(lst, cnt, None, self.obs, self.base_t)
            obs_stage = RecordRows(self.obs, self.base_t, self.C)
        elif dual:
            self.eng.observe_list(lst, cnt, obs_stage, self.obs, self.base_t)
        else:
            self.eng.observe_list(lst, cnt, obs_stage)
        if im:
            env = th.where((self.ar_c < cnt) & (lst >= 0), lst, self.N).long()
            self.eng.add_intrinsic_reward(self.obs, self.rec_of_env[env], obs_stage, lst, cnt, out["reward"])
        noise = log_std = None
        if self.fused and self.policy_parts_fn is not None:
            actions, log_std, values = self.policy_parts_fn(obs_stage)                 # `actions` is the mean here
            # (the synthetic stream draws its actions inside the recorder kernel: no noise tensor, one launch less per tick)
            noise = th.randn_like(actions) if self.rng_seed is None else None; log_probs = values
        else:
            actions, values, log_probs = self.policy_fn(obs_stage)
        rows = None
        if not (dual and self.fused):
            rows = self.base_t + self.ar_c                             # record ids of this tick
        if not dual:
            self.obs.index_copy_(0, rows, obs_stage)
        if self.fused:
            self._fused_tick(out, lst, cnt, slot_act, p, actions, values, log_probs, noise, log_std)
        else:
            self._torch_tick(out, lst, cnt, slot_act, rows, actions, values, log_probs)
        self.base_t += self.C

    def _tick_pipelined(self):
        p = self.total_ticks % 2
        main = th.cuda.current_stream(self.dev)
        if self.ev_side[p] is not None:
            main.wait_event(self.ev_side[p])                           # the actions this launch starts envs with are ready
        out = self.eng.advance(self.slot_act2[p], self.S, self.lst2[p], self.cnt2[p], lag=2)
        ev = th.cuda.Event(); ev.record(main)
        with th.cuda.stream(self.side):
            self.side.wait_event(ev)
            if self.use_graph and self._side_graph[p] is None and self.total_ticks >= 2 * self.graph_after:
                from .gemm_choices import recorded_gemm_choices
                g = th.cuda.CUDAGraph()
                with recorded_gemm_choices(), th.cuda.graph(g, stream=self.side, capture_error_mode="thread_local"):
                    self._decide(out, self.lst2[p], self.cnt2[p], self.obs_stage2[p], self.slot_act2[p], p)
                self._side_graph[p] = g
            if self._side_graph[p] is not None:
                self._side_graph[p].replay()
            elif self.use_graph:
                from .gemm_choices import recorded_gemm_choices
                with recorded_gemm_choices():                         # (the eager decisions before the capture: see _tick)
                    self._decide(out, self.lst2[p], self.cnt2[p], self.obs_stage2[p], self.slot_act2[p], p)
            else:
                self._decide(out, self.lst2[p], self.cnt2[p], self.obs_stage2[p], self.slot_act2[p], p)
            e2 = th.cuda.Event(); e2.record(self.side)
            self.ev_side[p] = e2
        self.tick += 1; self.total_ticks += 1

    def _torch_tick(self, out, lst, cnt, slot_act, rows, actions, values, log_probs):
        N, C, R = self.N, self.C, self.R
        valid = (self.ar_c < cnt) & (lst >= 0)                         # [C]; negative entries are holes (lists merged from several batches)
        env = th.where(valid, lst, N).long()                           # dump env N for empty rows
        env_c = env.clamp(max=N - 1)
        slot_act.copy_(th.max(th.min(actions, self.high), self.low))
        # close the previous decision of every listed env
        prev = self.rec_of_env[env]
        had = valid & (prev >= 0)
        prev_m = th.where(had, prev, R)
        rew = out["reward"][env_c].float(); dn = out["done"][env_c].float()
        self.rewards[prev_m] = rew; self.dones[prev_m] = dn
        self.next_rec[prev_m] = rows
        self.completed.index_fill_(0, prev_m, True)
        self.n_completed += had.sum()
        if "n_substeps" in out:
            self.substeps_total += (out["n_substeps"][env_c] * had).sum()
        # episode bookkeeping
        self.ep_ret[env] += th.where(had, rew, th.zeros_like(rew)); self.ep_len[env] += had.float()
        fin = had & (dn > 0)
        self.ep_ret_sum += (self.ep_ret[env] * fin).sum(); self.ep_len_sum += (self.ep_len[env] * fin).sum(); self.ep_count += fin.sum()
        keep = (~fin).float()
        self.ep_ret[env] *= keep; self.ep_len[env] *= keep
        # open the new decision
        self.actions.index_copy_(0, rows, actions); self.log_probs.index_copy_(0, rows, log_probs); self.values.index_copy_(0, rows, values)
        self.is_rec.index_copy_(0, rows, valid)
        self.completed.index_fill_(0, rows, False); self.next_rec.index_fill_(0, rows, -1)
        self.prev_rec.index_copy_(0, rows, th.where(had, prev, th.full_like(prev, -1)))
        self.rec_env.index_copy_(0, rows, th.where(valid, env, th.full_like(env, -1)))
        self.rec_of_env[env] = rows

    def _fused_tick(self, out, lst, cnt, slot_act, p, actions, values, log_probs, noise=None, log_std=None):
        from .. import engine as E
        import ctypes as C
        # the merged heads leave [mean | value] as columns of one [C, A + 1] fp32 matrix: read in place through row strides, no gathered copies
        def rows_of(t, inner):
            if t.dtype == th.float32 and t.dim() == inner + 1 and t.stride(0) > 0 and (inner == 0 or t.stride(1) == 1):
                return t, int(t.stride(0))
            t = t.float().contiguous()
            return t, 0
        actions, mstride = rows_of(actions, 1); values, vstride = rows_of(values, 0)
        if log_probs is not values:
            log_probs = log_probs.float().contiguous()
        if self._targs[p] is None:
            ptr = lambda t: t.data_ptr()
            self._targs[p] = E.RolloutTickC(
                n_envs=self.N, capacity=self.C, action_dim=self.A, n_records=self.R, ready_list=ptr(lst), ready_count=ptr(cnt), base=ptr(self.base_t),
                reward=ptr(out["reward"]), done=ptr(out["done"]), n_substeps=ptr(out["n_substeps"]) if "n_substeps" in out else None,
                low=ptr(self.low), high=ptr(self.high), slot_actions=ptr(slot_act), rec_of_env=ptr(self.rec_of_env), rewards=ptr(self.rewards),
                dones=ptr(self.dones), next_rec=ptr(self.next_rec), prev_rec=ptr(self.prev_rec), rec_env=ptr(self.rec_env), completed=ptr(self.completed),
                is_rec=ptr(self.is_rec), actions_buf=ptr(self.actions), log_probs_buf=ptr(self.log_probs), values_buf=ptr(self.values),
                n_completed=ptr(self.n_completed), substeps_total=ptr(self.substeps_total), ep_ret=ptr(self.ep_ret), ep_len=ptr(self.ep_len),
                ep_ret_sum=ptr(self.ep_ret_sum), ep_len_sum=ptr(self.ep_len_sum), ep_count=ptr(self.ep_count))
            assert out["reward"].dtype == th.float32 and out["done"].dtype == th.uint8
        a = self._targs[p]
        a.actions, a.values, a.log_probs = actions.data_ptr(), values.data_ptr(), log_probs.data_ptr()
        a.mean_stride, a.value_stride = mstride, vstride
        if log_std is not None:
            log_std = log_std.float().contiguous()
            noise = noise.float().contiguous() if noise is not None else None
            a.log_std = log_std.data_ptr()
            a.noise = noise.data_ptr() if noise is not None else None
        else:
            a.noise, a.log_std = None, None
        self._tick_keep = (actions, values, log_probs, noise, log_std)          # (alive until the launch is enqueued)
        if self.rng_seed is not None and log_std is not None:
            a.rng_count, a.rng_seed = self.rng_count.data_ptr(), self.rng_seed
        else:
            a.rng_count, a.rng_seed = None, 0
        stream = C.c_void_p(th.cuda.current_stream(self.dev).cuda_stream)
        if E.lib().grip_rollout_tick(C.byref(a), stream) != 0:
            raise E.GripError("grip_rollout_tick failed")

    def _tick(self):
        if self.pipeline:
            return self._tick_pipelined()
        use_graph = self.use_graph and self.dev.type == "cuda"
        if use_graph and self._graph is None and self.total_ticks >= self.graph_after:
            self._capture()
        # every `eager_every`-th tick runs outside the graph so that the engine can time its kernel with events
        if use_graph and self._graph is not None and (self.total_ticks % self.eager_every) != 0:
            self._graph.replay()
        elif use_graph and self._graph is None:
            # the eager ticks BEFORE the capture run with the recorded GEMM choices switched on, as the capture will: TunableOp's first look at a shape
            # (table lookup, library queries) is not allowed inside a stream capture, so every shape of the tick has to have been seen by then
            from .gemm_choices import recorded_gemm_choices
            with recorded_gemm_choices():
                self._tick_body()
        else:
            self._tick_body()
        self.tick += 1; self.total_ticks += 1

    def _capture(self):
        """Capture one tick. Capturing records the launches without running them, so the state is untouched. If the runtime
        refuses the capture the rollout goes on eagerly (slower, same results)."""
        th.cuda.synchronize(self.dev)
        try:
            from .gemm_choices import recorded_gemm_choices
            g = th.cuda.CUDAGraph()
            with recorded_gemm_choices(), th.cuda.graph(g, capture_error_mode="thread_local"):
                self._tick_body()
            self._graph = g
        except Exception as ex:                                   # noqa: BLE001 -- any capture failure means "no graph"
            import warnings
            warnings.warn(f"hipGraph capture of the rollout tick failed ({ex}); continuing without graphs")
            th.cuda.synchronize(self.dev)
            self.use_graph = False; self._graph = None

    # ------------------------------------------------------------------ rollout
    @property
    def carry0(self):
        return self.win * self.W

    @property
    def tick0(self):
        return self.win * self.W + self.N

    def window_rows(self):
        """Record rows of the current rollout: its carry rows, then the tick blocks written so far."""
        return th.arange(self.carry0, self.tick0 + self.tick * self.C, device=self.dev)

    def _begin(self):
        """Switch to the other record window, carry the in-flight decisions (one per env at most) into its first N rows and
        clear the rest of it. The previous window is left untouched (an update may still be reading it)."""
        N = self.N
        infl = self.rec_of_env[:N]
        has = infl >= 0
        src = th.where(has, infl, self.R)
        ob = self.obs[src]; ac = self.actions[src]; lp = self.log_probs[src]; va = self.values[src]
        self.win ^= 1
        c0, w1 = self.carry0, self.carry0 + self.W
        self.obs[c0:c0 + N] = ob; self.actions[c0:c0 + N] = ac; self.log_probs[c0:c0 + N] = lp; self.values[c0:c0 + N] = va
        self.is_rec[c0:w1] = False; self.completed[c0:w1] = False; self.next_rec[c0:w1] = -1; self.prev_rec[c0:w1] = -1
        self.advantages[c0:w1] = 0; self.rec_env[c0:w1] = -1
        self.is_rec[c0:c0 + N] = has
        ar = th.arange(N, device=self.dev)
        self.rec_env[c0:c0 + N] = th.where(has, ar, th.full_like(ar, -1))
        self.rec_of_env[:N] = th.where(has, ar + c0, th.full_like(ar, -1))
        self.n_completed.zero_()
        self.base_t.fill_(self.tick0)
        self.tick = 0

    def collect(self, on_poll=None):
        """Run ticks until `target` transitions have completed (checked every `poll_every` ticks) or the tick budget is
        spent, then compute advantages. Returns the number of completed transitions (host int)."""
        if not self._started:
            self.eng.reset(); self._started = True
        self._begin()
        done_n = 0
        next_poll = self.poll_every
        while self.tick < self.max_ticks:
            self._tick()
            if self.tick >= next_poll or self.tick == self.max_ticks:
                if self.pipeline:
                    th.cuda.synchronize(self.dev)
                done_n = int(self.n_completed.item())          # the only host sync of the rollout loop
                if on_poll is not None and on_poll(done_n) is False:
                    break
                if done_n >= self.target:
                    break
                # poll again when the rollout should be about full (every tick near the end): a late poll costs a tick of
                # transitions nobody trains on, an early one a pipeline bubble
                rate = max(1.0, done_n / self.tick)
                next_poll = self.tick + max(1, min(self.poll_every, int((self.target - done_n) / rate)))
        else:
            done_n = int(self.n_completed.item())
        if self.pipeline:
            th.cuda.synchronize(self.dev)                      # side stream drained: records complete, safe to train
            done_n = int(self.n_completed.item())
        self._gae()
        self._retune(done_n)
        return done_n

    def _gae(self):
        if self.fused:
            from .. import engine as E
            import ctypes as C_
            if self.returns.data_ptr() == self.advantages.data_ptr() or not self.returns.is_contiguous():
                self.returns = th.zeros_like(self.advantages)
            stream = C_.c_void_p(th.cuda.current_stream(self.dev).cuda_stream)
            p = lambda t: C_.c_void_p(t.data_ptr())
            if E.lib().grip_rollout_gae(self.N, p(self.rec_of_env), p(self.prev_rec), p(self.rewards), p(self.dones), p(self.values),
                                        float(self.gamma), float(self.lam), p(self.advantages), p(self.returns), stream) != 0:
                raise E.GripError("grip_rollout_gae failed")
            return
        N, C, R = self.N, self.C, self.R
        g, gl = self.gamma, self.gamma * self.lam
        blocks = [(self.carry0, self.tick0)] + [(self.tick0 + k * C, self.tick0 + (k + 1) * C) for k in range(self.tick)]
        for lo, hi in reversed(blocks):
            nr = self.next_rec[lo:hi]
            comp = self.completed[lo:hi]
            nrc = th.where(comp, nr, th.full_like(nr, R))
            nonterm = 1.0 - self.dones[lo:hi]
            delta = self.rewards[lo:hi] + g * self.values[nrc] * nonterm - self.values[lo:hi]
            adv = delta + gl * nonterm * self.advantages[nrc]          # advantages of in-flight records stay 0
            self.advantages[lo:hi] = th.where(comp, adv, th.zeros_like(adv))
        self.returns = self.advantages + self.values

    def training_indices(self, generator=None):
        """Exactly `target` record ids (so that every rank runs the same number of minibatches): the completed records in
        time order, truncated; with replacement only if the tick budget ran out before the rollout filled."""
        lo, hi = self.carry0, self.carry0 + self.W
        idx = (self.completed[lo:hi] & self.is_rec[lo:hi]).nonzero().flatten() + lo
        if idx.numel() >= self.target:
            return idx[:self.target]
        if idx.numel() == 0:
            raise RuntimeError("async rollout finished without a single completed transition")
        extra = idx[th.randint(idx.numel(), (self.target - idx.numel(),), device=self.dev, generator=generator)]
        return th.cat([idx, extra])

    def stats(self):
        c = float(self.ep_count.item())
        return {"episodes": c, "ep_rew_mean": float(self.ep_ret_sum.item()) / max(c, 1.0), "ep_len_mean": float(self.ep_len_sum.item()) / max(c, 1.0),
                "ticks": self.total_ticks}
