"""development: event-driven model of the step launches of G slabs sharing the chip's 512 workgroup slots (2 per CU, four waves each),
to price scheduling policies before building them (round 5).  Durations follow profiles/r04_a_timeline_rule_4096.txt.
policy: none  -- a workgroup serves its own launch only and leaves when its four waves have nothing left (today's k_env_step_x)
        steal -- a wave with nothing left takes the next env of ANY published launch; completion of a slab = all its envs done"""
import heapq
import sys
import numpy as np

def draw(n, rng):
    slow = rng.random(n) < 0.10
    it = np.where(slow, 50 + np.minimum(rng.exponential(45.0, n), 400), 50 + rng.integers(0, 5, n))
    team = np.zeros(n, bool)
    big = it >= 150
    team |= big & (rng.random(n) < 0.59)
    team |= slow & ~big & (rng.random(n) < 0.15)
    dur = np.where(team, 1465 + 11.9 * (it - 50), 1719 + 15.2 * (it - 50)) * (1 + 0.02 * rng.standard_normal(n))
    key = dur.copy()
    miss = slow & ~team & (rng.random(n) < 0.35)   # slow envs the scheduler's key does not know about
    key[miss] = 1719 * (1 + 0.02 * rng.standard_normal(miss.sum()))
    return dur, key, team

def run(G, policy, steps=40, NWG=512, host_us=120.0, sched_us=90.0, seed=0, total=4096):
    rng = np.random.default_rng(seed)
    n = total // G
    # per slab: queues of the current epoch
    class S: pass
    slabs = []
    for g in range(G):
        s = S(); s.teamq = []; s.q = []; s.left = 0; s.epoch = 0; s.t_done = []; slabs.append(s)
    ev = []   # (time, kind, payload)
    cnt = 0
    free_wg = NWG
    pending_wg = []   # FIFO of (slab) workgroups waiting for a slot
    idle_waves = []   # workgroups (steal policy) are modelled as 4 independent wave slots that poll
    t = 0.0
    busy_slot_us = 0.0
    def publish(s, now):
        dur, key, team = draw(n, rng)
        order = np.argsort(-key)
        s.teamq = [dur[i] for i in order if team[i]]
        s.q = [dur[i] for i in order if not team[i]]
        s.left = n
        nwg = min(n // 4 + max(1, n // 16), NWG)
        for _ in range(nwg):
            pending_wg.append(s)
    # workgroup state machine: returns next event for a workgroup
    def pick_bundle(s_own):
        if s_own.q: return s_own, s_own.q.pop(0)
        if policy == "steal":
            for o in slabs:
                if o.q: return o, o.q.pop(0)
        return None, None
    def pick_team(s_own):
        if s_own.teamq: return s_own, s_own.teamq.pop(0)
        if policy == "steal":
            for o in slabs:
                if o.teamq: return o, o.teamq.pop(0)
        return None, None
    seq = 0
    def push(tt, kind, pl):
        nonlocal seq
        seq += 1
        heapq.heappush(ev, (tt, seq, kind, pl))
    for g, s in enumerate(slabs):
        push(g * 30.0, "publish", s)
    done_steps = 0
    t_first = None
    wgs = {}
    wid = 0
    def start_wgs(now):
        nonlocal free_wg, wid, busy_slot_us
        while free_wg > 0 and pending_wg:
            s = pending_wg.pop(0)
            free_wg -= 1
            wid += 1
            # team phase
            o, d = pick_team(s)
            if o is not None:
                busy_slot_us += 4 * d
                push(now + d, "team_done", (wid, s, o))
                wgs[wid] = 4
            else:
                begin_bundle(wid, s, now)
    def begin_bundle(w, s, now):
        nonlocal busy_slot_us, free_wg
        live = 0
        for k in range(4):
            o, d = pick_bundle(s)
            if o is not None:
                busy_slot_us += d
                push(now + d, "wave_done", (w, s, o))
                live += 1
        wgs[w] = live
        if live == 0:
            del wgs[w]
            free_wg += 1
    def env_done(o, now):
        o.left -= 1
        if o.left == 0:
            o.epoch += 1
            o.t_done.append(now)
            push(now + host_us + sched_us, "publish", o)
    while ev:
        now, _, kind, pl = heapq.heappop(ev)
        if kind == "publish":
            if pl.epoch >= steps: continue
            publish(pl, now)
            start_wgs(now)
        elif kind == "team_done":
            w, s, o = pl
            env_done(o, now)
            o2, d = pick_team(s)
            if o2 is not None:
                busy_slot_us += 4 * d
                push(now + d, "team_done", (w, s, o2))
            else:
                begin_bundle(w, s, now)
                start_wgs(now)
        elif kind == "wave_done":
            w, s, o = pl
            env_done(o, now)
            o2, d = pick_bundle(s)
            if o2 is not None:
                busy_slot_us += d
                push(now + d, "wave_done", (w, s, o2))
            else:
                wgs[w] -= 1
                if wgs[w] == 0:
                    del wgs[w]
                    free_wg += 1
                    start_wgs(now)
        t = now
    # throughput over the steady part: from every slab's 5th completion to its last
    t0 = max(s.t_done[4] for s in slabs); t1 = min(s.t_done[-1] for s in slabs)
    k = sum(sum(1 for x in s.t_done if t0 < x <= t1) for s in slabs)
    return k * n / (t1 - t0) * 1e6, busy_slot_us / (t * NWG * 4)

if __name__ == "__main__":
    for G in (4, 8, 16):
        for pol in ("none", "steal"):
            r = [run(G, pol, seed=sd) for sd in range(3)]
            print("G %2d %-5s  %.0f k env-steps/s   slot utilisation %.2f" % (G, pol, np.mean([x[0] for x in r]) / 1e3, np.mean([x[1] for x in r])))
