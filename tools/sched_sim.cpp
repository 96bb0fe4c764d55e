// sched_sim.cpp -- wave-schedule simulator for render_kernel's ballot scheduler, driven by real per-lane walk traces
// (tests/hostsim: hostsim_trace_file).  It replays the frame's path segments through 64-lane waves under a scheduling
// policy and adds up issue cycles with a small cost model, so that a policy can be priced before it is built:
//   policy 0: the shipped one (a shade phase when >= T lanes are not walking, else node / leaf steps by majority vote)
//   policy 1: two contexts per lane (one walking, one parked: a fresh ray waiting for the lane, or a finished one waiting
//             for the shade phase); lanes whose walk ends swap in their parked fresh ray at refill points
//   policy 2: postponed leaves (speculative traversal, Aila & Laine 2009): a lane that reaches a leaf keeps the leaf in a one-entry
//             slot and goes on with its node steps; leaf steps serve every lane that holds a leaf, and are voted for by the lanes
//             that cannot go on without one.  (The trace is the unspeculated walk: the node steps that a later hit would have
//             pruned are not added, so the figure is an upper bound of the gain.)
// Build: g++ -O2 -std=c++17 tools/sched_sim.cpp -o /tmp/sched_sim ; run: /tmp/sched_sim trace.bin policy T [R]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <vector>

struct Seg { uint32_t first, n; };                 // steps[first .. first + n)
struct Sample { uint32_t first, n; };              // segs
static std::vector<uint8_t> steps;                 // 1 = inner node, 0x10 + k = leaf with k tests
static std::vector<Seg> segs;
static std::vector<Sample> samples;                // 64 per pixel, pixel-major
static uint32_t SPP = 64, CHUNK = 2;

// cost model (issue cycles per wave)
static double C_A = 2130, C_VOTE = 45, C_NODE = 210, C_LEAF0 = 150, C_LEAFK = 130, C_SWAP = 150;
static int STEPS_PER_VOTE = 4;
static double C_ABSORB = 0; // policy 2: what keeping the postponed leaf adds to a node step (env SIM_ABSORB)
static int PEEL = 0; // study: the walk's first node step (the root node) done by the shade phase, whose cost grows by C_PEEL
static double C_PEEL = 110;

struct Ctx {
    int state = 0;               // 0 needs a unit, 1 ready for the shade phase (hit resolved / new sample), 2 walking, 3 fresh (a ray waiting to walk), 4 done
    uint32_t unit = 0, sample = 0, seg = 0, step = 0; // sample: index within the unit; seg: index within the sample; step: within the segment
    bool has_unit = false, started = false;
    int pend = -1;               // policy 2: tests of the postponed leaf, -1 = none
};
struct Queue { uint64_t next = 0, total = 0; };

static const Seg &cur_seg(const Ctx &c) { const Sample &s = samples[(size_t)(c.unit / (SPP / CHUNK)) * SPP + (c.unit % (SPP / CHUNK)) * CHUNK + c.sample]; return segs[s.first + c.seg]; }
static uint32_t nsegs(const Ctx &c) { return samples[(size_t)(c.unit / (SPP / CHUNK)) * SPP + (c.unit % (SPP / CHUNK)) * CHUNK + c.sample].n; }

// the shade phase for one context: ends the segment it was in, starts the next one (or the next sample / unit).
// returns with state 2/3-candidate (has steps to walk: caller decides walking vs fresh), 1 (ready again), 0 (needs a unit, none left now) or 4
static void shade(Ctx &c, Queue &q, uint64_t &blk_next, uint64_t &blk_end, bool &exhausted)
{
    for (;;) {
        if (c.has_unit && c.started) {
            // a segment has just been resolved
            if (c.seg + 1 < nsegs(c)) c.seg++;
            else { c.seg = 0; c.sample++; if (c.sample == CHUNK) { c.has_unit = false; } }
        }
        if (!c.has_unit) {
            if (blk_next == blk_end) { // a new block of 128 units
                if (q.next < q.total) { blk_next = q.next; blk_end = std::min(q.total, q.next + 128); q.next = blk_end; }
                else { exhausted = true; c.state = 4; return; }
            }
            c.unit = (uint32_t)blk_next++; c.has_unit = true; c.sample = 0; c.seg = 0;
        }
        c.started = true;
        c.step = 0;
        const Seg &s = cur_seg(c);
        if (PEEL && s.n && steps[s.first] == 1) c.step = 1; // the root node's step was done inside the shade phase
        c.state = c.step < s.n ? 2 : 1; // walks, or misses the root box: ready again at once
        return;
    }
}

int main(int argc, char **argv)
{
    if (argc < 4) { printf("usage: sched_sim trace.bin policy T [R] [nwaves]\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<uint8_t> raw;
    { uint8_t buf[1 << 16]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + n); }
    fclose(f);
    const int policy = atoi(argv[2]) % 10, T = atoi(argv[3]), R = argc > 4 ? atoi(argv[4]) : 8;
    if (getenv("SIM_ABSORB")) C_ABSORB = atof(getenv("SIM_ABSORB"));
    if (getenv("SIM_STEPS")) STEPS_PER_VOTE = atoi(getenv("SIM_STEPS"));
    PEEL = atoi(argv[2]) / 10; // policy 10: the shipped schedule with the root node's step peeled into the shade phase
    if (PEEL) C_A += C_PEEL;
    for (size_t i = 0; i < raw.size(); i++) {
        const uint8_t b = raw[i];
        if (b == 0xFD) continue;
        if (b == 0xFE) { samples.push_back({ (uint32_t)segs.size(), 0 }); continue; }
        if (b == 0xFF) { segs.push_back({ (uint32_t)steps.size(), 0 }); samples.back().n++; continue; }
        steps.push_back(b); segs.back().n++;
    }
    const uint64_t npix = samples.size() / SPP;
    Queue q; q.total = npix * (SPP / CHUNK);
    const int NW = argc > 5 ? atoi(argv[5]) : (int)std::max<uint64_t>(8, q.total / 12900);
    const int NC = policy == 1 ? 2 : 1;
    struct Wave { double t = 0; std::vector<Ctx> c; uint64_t blk_next = 0, blk_end = 0; bool exhausted = false, finished = false; };
    std::vector<Wave> waves(NW);
    for (auto &w : waves) w.c.assign(64 * NC, Ctx());
    double sumA = 0, sumN = 0, sumL = 0, sumV = 0, sumS = 0;
    uint64_t nA = 0, lanesA = 0, nN = 0, lanesN = 0, nL = 0, lanesL = 0, nS = 0;
    typedef std::pair<double, int> Ev;
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> pq;
    for (int w = 0; w < NW; w++) pq.push({ 0.0, w });
    double tmax = 0;
    while (!pq.empty()) {
        const int wi = pq.top().second; pq.pop();
        Wave &w = waves[wi];
        // lane l: active context c[l*NC], parked c[l*NC+1]
        auto act = [&](int l) -> Ctx & { return w.c[l * NC]; };
        auto park = [&](int l) -> Ctx & { return w.c[l * NC + (NC - 1)]; };
        int nwalk = 0, nready = 0, nrefill = 0;
        for (int l = 0; l < 64; l++) {
            const Ctx &a = act(l);
            if (a.state == 2) nwalk++;
            bool fin = a.state == 0 || a.state == 1;
            if (NC == 2) { const Ctx &p = park(l); fin = fin || p.state == 0 || p.state == 1; if (a.state != 2 && p.state == 3) nrefill++; }
            if (fin) nready++;
        }
        if (nwalk == 0 && nready == 0 && nrefill == 0) { w.finished = true; tmax = std::max(tmax, w.t); continue; }
        if (NC == 2 && nrefill > 0 && (nrefill >= R || nwalk == 0)) { // refill: finished lanes swap in their parked fresh ray
            for (int l = 0; l < 64; l++) if (act(l).state != 2 && park(l).state == 3) { std::swap(act(l), park(l)); act(l).state = 2; }
            w.t += C_SWAP; sumS += C_SWAP; nS++;
            pq.push({ w.t, wi });
            continue;
        }
        if (nready >= T || nwalk == 0) {
            // ---------------- shade phase: one finished context per lane
            int lanes = 0;
            for (int l = 0; l < 64; l++) {
                Ctx *c = nullptr;
                bool parked = false;
                if (act(l).state == 0 || act(l).state == 1) c = &act(l);
                else if (NC == 2 && (park(l).state == 0 || park(l).state == 1)) { c = &park(l); parked = true; }
                if (!c) continue;
                lanes++;
                shade(*c, q, w.blk_next, w.blk_end, w.exhausted);
                if (c->state == 2 && NC == 2) {
                    if (parked) c->state = 3;                                   // the lane is busy walking: the fresh ray waits
                }
            }
            w.t += C_A; sumA += C_A; nA++; lanesA += lanes;
        } else {
            // ---------------- walk phase: votes until enough lanes wait for the shade phase (or can refill)
            if (policy == 2) {
                for (;;) {
                    // absorb: a lane at a leaf with an empty slot takes the leaf along
                    int n_inner = 0, n_blocked = 0, n_pend = 0, nfin = 0;
                    for (int l = 0; l < 64; l++) {
                        Ctx &a = act(l);
                        if (a.state == 2) {
                            const Seg &sg = cur_seg(a);
                            if (a.pend < 0 && a.step < sg.n && steps[sg.first + a.step] != 1) { a.pend = steps[sg.first + a.step] - 0x10; a.step++; }
                            if (a.step == sg.n && a.pend < 0) a.state = 1;
                        }
                        if (a.state == 2) {
                            const Seg &sg = cur_seg(a);
                            const bool inner = a.step < sg.n && steps[sg.first + a.step] == 1;
                            if (inner) n_inner++;
                            if (a.pend >= 0) { n_pend++; if (!inner) n_blocked++; }
                        }
                        if (a.state == 0 || a.state == 1) nfin++;
                    }
                    if (n_inner + n_pend == 0 || nfin >= T) break;
                    w.t += C_VOTE; sumV += C_VOTE;
                    if (n_inner >= n_blocked && n_inner > 0) {
                        for (int k = 0; k < STEPS_PER_VOTE; k++) {
                            int lanes = 0;
                            for (int l = 0; l < 64; l++) {
                                Ctx &a = act(l);
                                if (a.state != 2) continue;
                                const Seg &sg = cur_seg(a);
                                if (a.step >= sg.n || steps[sg.first + a.step] != 1) continue;
                                lanes++;
                                a.step++;
                            }
                            if (!lanes) break;
                            w.t += C_NODE + C_ABSORB; sumN += C_NODE + C_ABSORB; nN++; lanesN += lanes;
                        }
                    } else {
                        int lanes = 0, maxk = 0;
                        for (int l = 0; l < 64; l++) {
                            Ctx &a = act(l);
                            if (a.state != 2 || a.pend < 0) continue;
                            lanes++; maxk = std::max(maxk, a.pend);
                            a.pend = -1;
                        }
                        const double cst = C_LEAF0 + C_LEAFK * maxk;
                        w.t += cst; sumL += cst; nL++; lanesL += lanes;
                    }
                }
                pq.push({ w.t, wi });
                continue;
            }
            for (;;) {
                int n_inner = 0, n_leaf = 0, nfin = 0, nref = 0;
                for (int l = 0; l < 64; l++) {
                    const Ctx &a = act(l);
                    if (a.state == 2) { if (steps[cur_seg(a).first + a.step] == 1) n_inner++; else n_leaf++; }
                    bool fin = a.state == 0 || a.state == 1;
                    if (NC == 2) { const Ctx &p = park(l); fin = fin || p.state == 0 || p.state == 1; if (a.state != 2 && p.state == 3) nref++; }
                    if (fin) nfin++;
                }
                if (n_inner + n_leaf == 0 || nfin >= T || (NC == 2 && nref >= R)) break;
                w.t += C_VOTE; sumV += C_VOTE;
                if (n_inner >= n_leaf) {
                    for (int k = 0; k < STEPS_PER_VOTE; k++) {
                        int lanes = 0;
                        for (int l = 0; l < 64; l++) {
                            Ctx &a = act(l);
                            if (a.state != 2) continue;
                            const Seg &s = cur_seg(a);
                            if (steps[s.first + a.step] != 1) continue;
                            lanes++;
                            if (++a.step == s.n) a.state = 1;
                        }
                        if (!lanes) break;
                        w.t += C_NODE; sumN += C_NODE; nN++; lanesN += lanes;
                    }
                } else {
                    int lanes = 0, maxk = 0;
                    for (int l = 0; l < 64; l++) {
                        Ctx &a = act(l);
                        if (a.state != 2) continue;
                        const Seg &s = cur_seg(a);
                        const uint8_t b = steps[s.first + a.step];
                        if (b == 1) continue;
                        lanes++; maxk = std::max(maxk, (int)b - 0x10);
                        if (++a.step == s.n) a.state = 1;
                    }
                    const double cst = C_LEAF0 + C_LEAFK * maxk;
                    w.t += cst; sumL += cst; nL++; lanesL += lanes;
                }
            }
        }
        pq.push({ w.t, wi });
    }
    const double total = sumA + sumN + sumL + sumV + sumS;
    printf("policy %d T %d R %d waves %d: total %.4g cycles (shade %.1f %%, node %.1f %%, leaf %.1f %%, votes %.1f %%, swaps %.1f %%); shade runs %llu x %.1f lanes, node steps %llu x %.1f lanes, "
           "leaf steps %llu x %.1f lanes; slowest wave / mean %.3f\n", policy, T, R, NW, total, 100 * sumA / total, 100 * sumN / total, 100 * sumL / total, 100 * sumV / total,
           100 * sumS / total, (unsigned long long)nA, nA ? (double)lanesA / nA : 0, (unsigned long long)nN, nN ? (double)lanesN / nN : 0, (unsigned long long)nL,
           nL ? (double)lanesL / nL : 0, tmax / (total / NW));
    return 0;
}
