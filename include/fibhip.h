/*
 * fibhip.h — C ABI of libfibhip.so: the MI355X (gfx950) explicit time-stepper for 2D cardiac
 * reaction-diffusion.
 *
 * What this boundary replaces.  The reference (siravan/fib_tf) has no FFI for its hot path:
 * the lower face of its stepper is `tf.Session.run(op)` on ops built by `define()`
 * (ionic.py:188-204).  Every entry point below names the reference call it stands in for.
 * The upper face (IonicModel / define() / run() / add_pace_op() / fire_op() / image()) is
 * kept in Python (fib_tf_amd/ionic.py ...) and binds these symbols through ctypes
 * (fib_tf_amd/_lib.py; INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions: plain C, opaque handle, every function returns 0 on success or a negative
 * FIBHIP_E* code; fibhip_last_error() returns a thread-local message for the last failure.
 * Host buffers are caller-owned float32, row-major; "slab" = SoA [nvar][height][width].
 * Work is enqueued asynchronously on the handle's HIP stream; only get/probe/sync/time block.
 * A handle is not re-entrant.  There is NO CPU fallback: without a HIP device create() fails.
 *
 * Variable order (index `var`), as the reference declares its state:
 *   FENTON4V : U V W S                                   (fenton.py:128-131)
 *   BR       : V C M H J D F XI                          (br.py:87-94)
 *   COURT    : V Na_i m h j K_i oa oi ua ui xr xs Ca_i d f f_Ca Ca_rel u v w Ca_up
 *                                                        (court.py:57-78)
 * Variable 0 is always the transmembrane potential (the only array the stencil touches).
 */
#ifndef FIBHIP_H
#define FIBHIP_H

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: these entry points are ALL it exports (no kernel handle, host stub or
 * template instance of one build can meet its namesake of another build of the same sources in one process). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define FIBHIP_ABI_VERSION 1

typedef struct fibhip_ctx *fibhip_t;

enum fibhip_model {
    FIBHIP_FENTON4V = 0, FIBHIP_BR = 1, FIBHIP_COURT = 2,
    FIBHIP_COURT_US = 3,  /* court_ultra.py with config['ultra_slow']: a 22nd array `_us_` (court_ultra.py:81-82,  */
                          /* 198-199,221-222,445-450); always single-rate (implies FIBHIP_ALLVARS)                */
    FIBHIP_CUSTOM = 4     /* a user-written IonicModel subclass in the reference's style, traced by               */
                          /* fib_tf_amd/traced.py and compiled into its own copy of this library                  */
                          /* (-DFIB_CUSTOM_MODEL_INC=...); the stock library rejects it                           */
};

enum fibhip_flags {
    FIBHIP_CHEBY   = 1u << 0, /* BR: Chebyshev gates, config['cheby'] (br.py:132-135); needs set_consts   */
    FIBHIP_SKIP    = 1u << 1, /* BR: multirate slow gates, config['skip'] (br.py:98-103)                  */
    FIBHIP_CHRONIC = 1u << 2, /* COURT: self.chronic (court.py:41,167-170)                                */
    FIBHIP_FAST    = 1u << 3, /* hardware-rate division/exp/tanh instead of the rounding-faithful forms;  */
                              /* looser parity tolerance, see DESIGN.md                                    */
    FIBHIP_ALLVARS = 1u << 4, /* COURT: every tick updates all 21 variables with dt (court_ultra.py:      */
                              /* 107-111,127-128) instead of the fast/slow split                           */
    FIBHIP_ZEROPAD = 1u << 6, /* Laplacian of fenton_simple.py / fenton_jit.py: a 3x3 convolution with zero padding        */
                              /* (fenton_simple.py:38-49) instead of IonicModel.laplace; single device only              */
    FIBHIP_HOLD    = 1u << 7, /* BR: the slow gates xi, j, d, f are not advanced at all: BeelerReuter.solve(state, 0),  */
                              /* the sub-steps 1..4 of a `skip` tick taken on their own (br.py:98-103,195-205)         */
    FIBHIP_ROW_INTERLEAVED = 1u << 5 /* device slab layout [height][nvar][width] instead of               */
                              /* [nvar][height][width]: the rows a row block exchanges with a neighbour    */
                              /* (all arrays) are then ONE contiguous block.  Host-side get/set_state keep  */
                              /* the planar [nvar][height][width] view.                                     */
};

enum fibhip_err {
    FIBHIP_OK = 0,
    FIBHIP_EINVAL = -1,   /* bad argument / call order                                   */
    FIBHIP_EHIP = -2,     /* a HIP runtime call failed (message has the HIP error text)  */
    FIBHIP_ENODEV = -3,   /* no usable HIP device                                        */
    FIBHIP_ENOMEM = -4
};

typedef struct fibhip_desc {
    int struct_size;     /* sizeof(fibhip_desc): ABI check                                                */
    int model;           /* enum fibhip_model                                                             */
    int height, width;   /* rows/cols of THIS handle's slab (for a row block: owned + ghost rows)         */
    double dt;           /* config['dt']   (fenton.py:159)                                                */
    double diff;         /* config['diff'] (fenton.py:161); diff*dt is formed in double, rounded once     */
    unsigned flags;      /* enum fibhip_flags                                                             */
    int device;          /* HIP device ordinal                                                            */
    int steps_per_tick;  /* sub-steps one tick advances; 0 = the reference's unroll factor               */
                         /* (Fenton 10 fenton.py:135-138, BR 5 br.py:98-107, Courtemanche 1 court.py:92)  */
    /* row-block domain decomposition (all 0 for a single device):                                       */
    int global_height;   /* rows of the whole grid; 0 = height                                            */
    int row_offset;      /* global row index of local row 0                                               */
    int ghost_top;       /* rows at the top of the slab that mirror the upper neighbour's rows            */
    int ghost_bottom;    /* same at the bottom; a ghost width must be >= steps_per_tick.  A width of      */
                         /* m * steps_per_tick makes the halo exchange due only every m-th tick           */
                         /* (fibhip_halo_due): the ticks in between also advance the ghost rows they      */
                         /* still need (communication-avoiding, redundant compute instead of messages)    */
    void *stream;        /* hipStream_t to enqueue on (e.g. the caller's torch stream); NULL = own stream */
    void *ext_slab[2];   /* optional caller-owned DEVICE slabs, each nvar*height*width floats             */
                         /* (so that the caller can hand them to RCCL); NULL = library allocates          */
    void *module;        /* model == FIBHIP_CUSTOM on the stock library: a fibhip_module_t (fibhip_module_load)  */
                         /* holding the traced model's kernels; NULL otherwise                            */
} fibhip_desc;

/* model facts, usable before create: number of state arrays / default steps per tick */
int fibhip_nvar(int model);
int fibhip_default_steps_per_tick(int model);
int fibhip_abi_version(void);
int fibhip_device_count(void);

/* == define(): tf.Variable creation + graph build (fenton.py:126-147, br.py:85-122, court.py:85-112) */
int fibhip_create(const fibhip_desc *desc, fibhip_t *out);
int fibhip_destroy(fibhip_t h);

/* == self.ϕ = tf.Variable(self.phase) (ionic.py:55-57).  phi: host [height*width] (local rows). Must
 * precede the first step.  NULL removes the phase field.                                               */
int fibhip_set_phase(fibhip_t h, const float *phi);

/* == tf.Variable(init) / define(state=...) (court.py:87-89).  var = -1: whole slab (restarts a row block's
 * exchange cycle: the ghost rows come with it); var >= 0: one array, the cycle position is kept.          */
int fibhip_set_state(fibhip_t h, int var, const float *src);

/* == Variable.eval() (fenton.py:152-153, ionic.py:226-229).  Blocks until preceding ticks are done.     */
int fibhip_get_state(fibhip_t h, int var, float *dst);
/* The same read-back without the staging copy: `dst` should be page-locked memory from fibhip_host_alloc (the Python
 * binding keeps a small pool of such buffers behind the arrays `eval()` / `image()` return: the reference driver reads
 * the potential back 100 times per simulated second, fenton.py:184-185).
 * Run-ahead: when the lengths of the caller's last series of ticks repeat (the same again, or with a period of up to four:
 * that very driver, also inside benchmark regions), this call launches the NEXT series before it returns — the frame
 * then travels inside that launch
 * (the device writes `dst` itself) — and fibhip_step hands those ticks out without launching; any other call on the
 * handle first restores exactly the state the caller has been told about (the launch is stopped at the tick the caller
 * has reached; if a tile is past it already, those ticks are recomputed and the rest cancelled).
 * Invisible except in time; FIBHIP_AHEAD=0 switches it off.  Never on caller-owned slabs (fibhip_desc.ext_slab) and never
 * again once fibhip_state_ptr has handed out a raw pointer: such a caller may write the state between two calls.        */
int fibhip_get_state_direct(fibhip_t h, int var, float *dst);
int fibhip_host_alloc(size_t nbytes, void **out);
int fibhip_host_free(void *p);

/* == constants baked into the graph at define time.  BR + FIBHIP_CHEBY: the 12x9 float32 table `d` of
 * br.py:327 in row order m_inf,h_inf,m_tau,h_tau,xi_inf,j_inf,d_inf,f_inf,xi_tau,j_tau,d_tau,f_tau
 * (br.py:223-240).                                                                                      */
int fibhip_set_consts(fibhip_t h, const float *tbl, int n);

/* == nticks x sess.run(self._ode_op) (ionic.py:202-203).  Asynchronous: an ENQUEUE.  The library may hold ticks back
 * until a launch is worth issuing — the last tick of a call (fibhip_step_slow fuses with it), up to three ticks
 * (Courtemanche, fast policy) or up to 32 (Fenton / Beeler-Reuter on grids whose tiles are all resident at once: one
 * launch loops over them, see fibhip_ticks_per_launch) — and any call on the handle that observes or changes the state
 * launches what is held first: invisible to the caller except that work may be enqueued a few calls later — or
 * EARLIER: a caller whose last series of ticks (between two observations) had n ticks gets the next n launched at the
 * first of them and handed out call by call; if it then stops after fewer, the running launch is told to stop at that
 * tick (or, too late for that, those ticks are recomputed).  fibhip_spec_stats counts both outcomes.                 */
int fibhip_step(fibhip_t h, int nticks);
/* == the loop bounds of IonicModel.run() (ionic.py:199-206: `for i in range(samples)`, a frame every dt_per_plot ticks): the
 * caller DECLARES that it will ask for the next `nticks` ticks as one series — no observation or change of the state in
 * between; one observation may still come first (the frame that starts the series).  A declared series is launched at its
 * first tick, up to fibhip_ticks_per_launch at a time, and handed out call by call; nothing is guessed from the call
 * history (what fibhip_step does for callers that do not say).  A caller that breaks its word is served like one whose
 * predicted series was wrong: the running launch is stopped at the tick reached.  0 withdraws the declaration.
 * TensorFlow has no counterpart: sess.run is synchronous (ionic.py:202-204).                                          */
int fibhip_expect(fibhip_t h, int nticks);

/* == fire_op('slow') of Courtemanche (court.py:103,615-617): re-evaluates solve on the current state and
 * assigns the 17 slow variables.                                                                        */
int fibhip_step_slow(fibhip_t h);
/* == sess.run(op) of any further assign group of a traced model (mode >= 1; FIBHIP_COURT: mode 1 is the 'slow'
 * op): the model is re-evaluated on the current state without the Laplacian and the mode's variables assigned  */
int fibhip_step_mode(fibhip_t h, int mode);

/* == fire_op(name) of an add_pace_op (ionic.py:144-169): pot = max(pot, s) with s = v inside the GLOBAL
 * rectangle rows [r0,r1) x cols [c0,c1) and min_v outside.                                              */
int fibhip_pace(fibhip_t h, int r0, int r1, int c0, int c1, float v, float min_v);

/* == the 'trend' probe (court.py:107-111): one value at LOCAL (row, col).  Blocks.                       */
int fibhip_probe(fibhip_t h, int var, int row, int col, float *out);

int fibhip_sync(fibhip_t h);

/* Runs `nticks` ticks bracketed by HIP events on the handle's stream and returns the elapsed
 * milliseconds and the number of kernel launches in between (for the roofline line of bench.py).        */
int fibhip_time_steps(fibhip_t h, int nticks, float *elapsed_ms, int *launches);
/* The same bracket around whatever the caller enqueues between the two calls (ticks, 'slow' ops, pacing):
 * the real tick mix of a reference driver loop (court.py:612-617) timed on the kernels' own stream.      */
int fibhip_time_begin(fibhip_t h);
int fibhip_time_end(fibhip_t h, float *elapsed_ms, int *launches);

/* ---- row-block decomposition plumbing (multi-GPU; the halo exchange itself is the caller's RCCL) ----
 * One tick = step_edges (the tiles that produce the rows a neighbour needs; main stream) +
 * step_interior (everything else; second stream, concurrent with the caller's halo exchange) +
 * step_commit (join the streams, flip the ping-pong buffers).  fibhip_step == the three in sequence.   */
int fibhip_step_edges(fibhip_t h);
int fibhip_step_interior(fibhip_t h);
int fibhip_step_commit(fibhip_t h);
/* Which of the two slabs currently holds `var` AFTER the last commit (0/1), and its device address.
 * Between step_edges and step_commit, `fibhip_next_ptr` gives the buffer being written.                 */
int fibhip_state_ptr(fibhip_t h, int var, void **dev_ptr);
int fibhip_next_ptr(fibhip_t h, int var, void **dev_ptr);
/* number of state arrays whose ghost rows must be refreshed after each tick (all of them when a tick
 * fuses several sub-steps, only the potential when steps_per_tick == 1)                                 */
int fibhip_halo_vars(fibhip_t h);
/* 1 if the tick that is open (or, between ticks, the next one) ends an exchange cycle: the caller must then
 * refresh ALL ghost rows of the fibhip_halo_vars arrays between step_edges and step_commit; 0 otherwise
 * (step_edges launches everything, step_interior nothing, no exchange)                                    */
int fibhip_halo_due(fibhip_t h);

/* IonicModel's public building blocks as stand-alone array ops on HOST arrays [H*W] (unit-level parity):
 * op 0 = enforce_boundary(a)            ionic.py:107-113
 *    1 = laplace(a) (+ phase term if phi) ionic.py:44-60
 *    2 = phase_field(REFLECT-pad(a)), phi ionic.py:70-81
 *    3 = rush_larsen(a=g, b=g_inf, c=tau, dt)  ionic.py:115-123                                         */
int fibhip_unit_op(int device, int op, int height, int width, const float *a, const float *b, const float *c,
                   const float *phi, double dt, int fast, float *out);

/* Courtemanche.calc_inter(V, mod) (court.py:273-429; court_ultra.py:445-450 for the last two) on a HOST array
 * of n voltages: out[k*n + i] = k-th intermediate of V[i], k in the insertion order of the reference's dict:
 * d_infinity tau_d f_infinity tau_f tau_w w_infinity m_inf tau_m h_inf tau_h j_inf tau_j tau_oa oa_infinity
 * tau_oi oi_infinity tau_ua ua_infinity tau_ui ui_infinity tau_xr xr_infinity tau_xs xs_infinity g_Kur f_NaK
 * i_NaCaa i_NaCab i_K1a i_Kra us_infinity tau_us   (FIBHIP_COURT_NINTER = 32 rows)                        */
/* Direct halo exchange (opt-in, FIBTF_HALO=direct): the ghost rows move by ncclSend/ncclRecv issued by the library
 * itself on the handle's stream, one RCCL kernel per exchange, instead of through the caller's communication
 * library.  RCCL is bound with dlopen: pass the path of the librccl the process uses (NULL if it is already loaded).
 *   fibhip_comm_open       once per process
 *   fibhip_comm_unique_id  on one rank; the caller broadcasts the 128 bytes
 *   fibhip_comm_check      every local reason fibhip_comm_init could refuse, without the collective: run it on every
 *                          rank and agree on the outcome BEFORE any rank enters fibhip_comm_init
 *   fibhip_comm_init       collective: every rank of the group, same id
 *   fibhip_comm_exchange   on an exchange tick (fibhip_halo_due), between step_edges and step_commit; -1 = no
 *                          neighbour on that side.  Needs FIBHIP_ROW_INTERLEAVED (one block per message).         */
typedef struct fibhip_halo_msg {
    long long offset;   /* floats from the base of the slab fibhip_halo_plan names */
    long long count;    /* floats */
    int peer;           /* rank of the neighbour */
    int send;           /* 1 = my rows go out, 0 = the neighbour's rows come in (my ghost rows) */
} fibhip_halo_msg;
/* the (at most 4) messages of one halo exchange of the open tick, in posting order; returns their number.
 * *slab_index = which of the two slabs (ext_slab[i]) they refer to.                                              */
int fibhip_halo_plan(fibhip_t h, int up_rank, int down_rank, fibhip_halo_msg *out4, int *slab_index);
int fibhip_comm_open(const char *librccl_path);
int fibhip_comm_unique_id(char *out128);
int fibhip_comm_check(fibhip_t h, int rank, int nranks);   /* the local (non-collective) checks of comm_init */
int fibhip_comm_init(fibhip_t h, const char *id128, int rank, int nranks);
int fibhip_comm_exchange(fibhip_t h, int up_rank, int down_rank);
int fibhip_comm_free(fibhip_t h);

/* Run-time modules: a traced model (FIBHIP_CUSTOM) without a compiler on the box and without a library of its own.
 * The caller compiles the model's DEVICE code in-process (hiprtc: the generated `struct Custom` + csrc/kernels.hpp,
 * one name expression per kernel below) and hands the code object over; the stock library loads it and drives its
 * kernels with the same host logic as the built-in models.  `fibhip_desc.module` then selects it at fibhip_create.  */
typedef struct fibhip_module_kernel {
    const char *symbol;  /* lowered (mangled) name of the kernel in the code object                           */
    int kind;            /* 0 tick_kernel<Custom,P,MODE,K,TX,TY,NT,PHASE>, 1 strip_kernel<...,K,TX,TY,R,PHASE>,  */
                         /* 2 pointwise_kernel<Custom,P,MODE> (an assign group fired by fibhip_step_mode)       */
    int mode, fast, phase;
    int K, TX, TY, NT;   /* NT: threads per workgroup (tick) or -R (strip), as csrc/fibhip.hip lists its own    */
} fibhip_module_kernel;
typedef struct fibhip_module_desc {
    int struct_size;     /* sizeof(fibhip_module_desc)                                                        */
    int nvar;            /* Custom::NVAR                                                                      */
    int steps_per_tick;  /* Custom::DEFAULT_STEPS                                                             */
    int nmodes;          /* Custom::NMODES (assign groups; mode 0 = the tick op)                              */
    unsigned masks[8];   /* Custom::mask(mode): which arrays each group assigns                               */
    int consts_bytes;    /* sizeof(Custom::Consts)                                                            */
    int K, TX, TY, R, TYB, K2, TX2, TY2, R2;   /* FIB_CUSTOM_* plan hints of the generated header              */
    int nkernels;
    const fibhip_module_kernel *kernels;
} fibhip_module_desc;
typedef struct fibhip_module *fibhip_module_t;
int fibhip_module_load(int device, const void *code, size_t nbytes, const fibhip_module_desc *d, fibhip_module_t *out);
int fibhip_module_unload(fibhip_module_t m);   /* after every handle created on it has been destroyed */

/* measurement aid: best-of-`reps` rate of a plain device-to-device streaming copy of nbytes (read + written bytes
 * per second, GB/s) — the achievable-bandwidth yardstick printed next to the roofline peak                  */
int fibhip_copy_bandwidth(int device, size_t nbytes, int reps, float *gb_per_s);

/* Launches one empty kernel of THIS build of the library on `device`, so that its code object is the first one the HIP
 * runtime brings up.  fib_tf_amd calls it on the stock library before it uses any other build (a Beeler-Reuter table
 * baked in, a traced model compiled in): under rocprofv3 (ROCm 7.2) the first launch from the stock library AFTER
 * another build's kernels have run dies inside libamdhip64's launch path; the other order is fine.   */
int fibhip_warm(int device);

#define FIBHIP_COURT_NINTER 32
int fibhip_court_inter(int device, int n, const float *V, int fast, float *out);

/* introspection for DESIGN/bench: sub-steps fused per launch and launches per tick                      */
int fibhip_launch_plan(fibhip_t h, int *fused_steps, int *launches_per_tick);
/* ... and the tile of the first launch: width, height, and rows per wave (strip kernels) or -threads per workgroup
 * (flat tile kernels)                                                                                              */
int fibhip_plan_tile(fibhip_t h, int *tile_w, int *tile_h, int *rows_per_wave);
/* Consecutive ticks one launch can cover (1 = every tick is its own launch or launches).  Courtemanche under
 * FIBHIP_FAST on one device returns 3: fibhip_step() accepts ticks and launches them three at a time, temporally
 * blocked; whatever has been accepted but not launched is launched by the next call that observes or changes the
 * state (get/set_state, probe, pace, sync, step_slow ...), so no caller can see the difference.
 * Fenton / Beeler-Reuter grids whose tiles are all resident on the device at once (tiles <= compute units; one device,
 * planar slab) return FIBHIP_MT_MAX (default 32): one launch then loops over up to that many ticks, its tiles re-reading
 * only the rim of their compute box from their neighbours between two ticks (no TensorFlow counterpart: ionic.py:202-204
 * issues one sess.run per tick).  A launch goes out as soon as the device would otherwise idle (see fibhip_step in
 * csrc/fibhip.hip); FIBHIP_MT=0 switches the mode off.  A tile of such a launch that waits 2 s for a neighbour gives up:
 * the next synchronising call restores the state that launch started from and goes on with one launch per tick
 * (fibhip_fallbacks).                                                                                             */
int fibhip_ticks_per_launch(fibhip_t h);
/* counters since fibhip_create: out[0] launches of any kernel, out[1] ticks advanced, out[2] multi-tick launches,
 * out[3] ticks those advanced (profiling scripts turn per-launch hardware counters into per-tick figures with them)  */
int fibhip_launch_stats(fibhip_t h, long long out[4]);
/* series of ticks that were launched ahead of the caller (fibhip_step: a whole series at its first tick when the last
 * series had that length; fibhip_get_state_direct: the next series before the frame is waited for) and that the caller
 * then cut short: out[0] launches that were stopped at the tick the caller had reached (nothing computed twice), out[1]
 * launches whose handed-out ticks had to be recomputed.  No TensorFlow counterpart (ionic.py:202-204 is synchronous).   */
int fibhip_spec_stats(fibhip_t h, long long out[2]);
/* Multi-tick launches that gave up waiting (a tile waited its full bound for a neighbour: not every workgroup was resident —
 * a CU mask, another process holding the device) and were RECOVERED: the handle went back to the state the launch had
 * started from (a launch writes the other slab only), switched multi-tick launches off for good and recomputed the lost
 * ticks one launch per tick — bit-identical, only slower.  out[0] such launches, out[1] ticks recomputed.  Non-zero out[0]
 * is worth a warning (fib_tf_amd prints one).  No TensorFlow counterpart (ionic.py:202-204 cannot fail this way).
 * fibhip_set_mt_wait_ms: the bound, per handle (default 2000; FIBHIP_MT_WAIT_MS presets it).                          */
int fibhip_fallbacks(fibhip_t h, long long out[2]);
int fibhip_set_mt_wait_ms(fibhip_t h, int ms);

/* Timeline of the launches of a tick (ionic.py:231-241 traces one sess.run with TensorFlow's timeline and writes a
 * Chrome trace): between trace_begin and trace_end every kernel launch of the handle is bracketed by a pair of HIP
 * events on its stream and nothing is deferred (one launch = one tick); trace_end waits for the stream and returns
 * the events: kernel family and shape, start relative to the first event and duration in microseconds.            */
typedef struct fibhip_trace_event {
    char name[96];            /* e.g. "strip_kernel<K=10, tile 44x25, 3 rows per wave>" */
    double start_us, dur_us;
    int K, tile_w, tile_h, rows_per_wave, ticks;
} fibhip_trace_event;
int fibhip_trace_begin(fibhip_t h);
int fibhip_trace_end(fibhip_t h, fibhip_trace_event *out, int max_events);   /* returns the number of events traced; the first */
                                                                              /* max_events of them are written (more = truncated) */

const char *fibhip_last_error(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FIBHIP_H */
