"""ONE configuration object for every choice of kernel form the package makes (round 5).

Until round 4 eleven environment variables were read at call time, scattered over nets.py, ops.py, finetune.py and solver.py, and
bench.py had to mutate `os.environ` (and patch packed weights) to pin a form.  Now:

    from adaptivepnp_sci_amd import config
    cfg = config.Config(precision='f16x3', streams=1)          # an explicit, immutable description of the forms
    with config.use(cfg):                                      # ... for everything constructed / stepped by this thread in here
        run = AdmmRun(y, Phi, 'ffdnet_color', True, model=net)
    run = AdmmRun(..., config=cfg)                             # ... or handed to a solve, which keeps it for all its steps
    with config.use(wino_f4=False): ...                        # field overrides on top of the current configuration

`current()` is the only place the environment is consulted: the process default is `Config.from_env()`, built on first use and
rebuilt only when one of the variables below has changed since (so `SCIPNP_CONV_PRECISION=f16x3 python ...` and a test's
`monkeypatch.setenv` keep working); an explicit configuration -- `use(...)`, `set_default(...)`, `AdmmRun(config=...)` -- always wins
over it.  The C boundary mirrors the same choices in its argument blocks: `scipnp_twostage_ffdnet_args.conv_form`
(include/scipnp.h) names the arithmetic the block's pointers must provide, `side_stream` the second stream, `defer_state` the
deferred ADMM-TV dual update; the library itself reads no environment variable on the product path.

field          environment variable                         values
precision      SCIPNP_CONV_PRECISION (SCIPNP_FFDNET_PRECISION)  'f32' (default: exact fp32 products on the fp32 MFMA) | 'f16x3'
f32_form       SCIPNP_F32_CONV                              'winograd' (default) | 'direct'
wino_f4        SCIPNP_WINO_F4                               True (default: F(4x4,3x3) where the layer shape has it) | False: F(2x2,3x3)
f32_wgrad      SCIPNP_F32_WGRAD                             'f4' (default: FFDNet trainer's weight gradient in the F(4x4) domain) | 'f2'
streams        SCIPNP_STREAMS                               1..8 (default 2): half-batches of a network pass on side streams
hipgraph       SCIPNP_HIPGRAPH                              False (default) | True: ADMM-TV schedules replayed as a hipGraph
tv_defer       SCIPNP_TV_DEFER                              True (default: two-launch ADMM-TV iteration) | False
resident_trainer SCIPNP_RESIDENT_TRAINER                    True (default: an update_=True FFDNet solve builds its finetune trainer with the engine and
                                                            keeps it -- activation stash, F(4x4) slab workspaces, Adam state: ~1.8 GB per 256x256x16
                                                            tile, > 3 GB at 512x512x8 -- for the solve's lifetime) | False: built at the event, freed after it
wgrad_slabs    SCIPNP_WGRAD_SLABS                           None (default: one persistent workgroup per CU) | int
(SCIPNP_LIB -- another libscipnp.so -- and SCIPNP_KEEP_TORCH_THREADS are process-level switches of _lib.py, not forms.)
"""
import contextlib
import dataclasses
import os
import threading
from typing import Optional

_ENV = ('SCIPNP_CONV_PRECISION', 'SCIPNP_FFDNET_PRECISION', 'SCIPNP_F32_CONV', 'SCIPNP_WINO_F4', 'SCIPNP_F32_WGRAD', 'SCIPNP_STREAMS',
        'SCIPNP_HIPGRAPH', 'SCIPNP_TV_DEFER', 'SCIPNP_WGRAD_SLABS', 'SCIPNP_RESIDENT_TRAINER')


@dataclasses.dataclass(frozen=True)
class Config:
    precision: str = 'f32'
    f32_form: str = 'winograd'
    wino_f4: bool = True
    f32_wgrad: str = 'f4'
    streams: int = 2
    hipgraph: bool = False
    tv_defer: bool = True
    wgrad_slabs: Optional[int] = None
    resident_trainer: bool = True

    def __post_init__(self):
        if self.precision not in ('f32', 'f16x3'):
            raise ValueError("SCIPNP_CONV_PRECISION must be 'f32' or 'f16x3'")
        if self.f32_form not in ('winograd', 'direct'):
            raise ValueError("SCIPNP_F32_CONV must be 'winograd' or 'direct'")
        if self.f32_wgrad not in ('f2', 'f4'):
            raise ValueError("SCIPNP_F32_WGRAD must be 'f2' or 'f4'")
        if not 1 <= int(self.streams) <= 8:
            raise ValueError('SCIPNP_STREAMS must be 1..8')
        if self.wgrad_slabs is not None and int(self.wgrad_slabs) < 1:
            raise ValueError('SCIPNP_WGRAD_SLABS must be a positive integer')

    def replace(self, **fields):
        return dataclasses.replace(self, **fields)

    @classmethod
    def from_env(cls, environ=None):
        e = os.environ if environ is None else environ
        slabs = e.get('SCIPNP_WGRAD_SLABS')
        return cls(precision=e.get('SCIPNP_CONV_PRECISION', e.get('SCIPNP_FFDNET_PRECISION', 'f32')),
                   f32_form=e.get('SCIPNP_F32_CONV', 'winograd'),
                   wino_f4=e.get('SCIPNP_WINO_F4', '1') != '0',
                   f32_wgrad=e.get('SCIPNP_F32_WGRAD', 'f4').lower(),
                   streams=int(e.get('SCIPNP_STREAMS', '2')),
                   hipgraph=e.get('SCIPNP_HIPGRAPH', '0') == '1',
                   tv_defer=e.get('SCIPNP_TV_DEFER', '1') != '0',
                   wgrad_slabs=int(slabs) if slabs else None,
                   resident_trainer=e.get('SCIPNP_RESIDENT_TRAINER', '1') != '0')


_lock = threading.Lock()
_default = None                  # (environment snapshot, Config) of the process default, or ('explicit', Config) after set_default
_tls = threading.local()         # .stack: this thread's use(...) overrides


def _env_key():
    e = os.environ
    return tuple(e.get(k) for k in _ENV)


def default():
    """the process default: set_default()'s configuration, else Config.from_env() (rebuilt only when the environment changed)"""
    global _default
    d = _default
    if d is not None and d[0] == 'explicit':
        return d[1]
    key = _env_key()
    if d is None or d[0] != key:
        with _lock:
            _default = d = (key, Config.from_env())
    return d[1]


def set_default(cfg):
    """make `cfg` the process default (None: back to the environment)"""
    global _default
    with _lock:
        _default = None if cfg is None else ('explicit', cfg)


def current():
    """the configuration in force for the calling thread: the innermost use(...) / solve scope, else the process default"""
    st = getattr(_tls, 'stack', None)
    return st[-1][2] if st else default()


def _push(base, over):
    st = getattr(_tls, 'stack', None)
    if st is None:
        st = _tls.stack = []
    st.append((base, over, base.replace(**over) if over else base))
    return st


@contextlib.contextmanager
def use(cfg=None, **fields):
    """In force for this thread inside the block: `cfg` (default: what is in force now) with `fields` replaced.  Field overrides
    given WITHOUT a cfg also reach into the solves stepped inside the block (`with config.use(streams=1): run.step(...)` runs that
    step on one stream although the run keeps its own configuration): explicit field overrides > a solve's configuration > a
    full configuration of an outer block > the process default.  Exception: a field the solve's constructor was given explicitly
    (AdmmRun(config=...) pins every field, conv_precision= the precision) stays what the solve was built with -- its buffers and
    packed weights exist in that form -- unless it is one of SCHEDULING_FIELDS (`streams`), which never change a result."""
    st = getattr(_tls, 'stack', None)
    if cfg is not None:
        base, over = cfg, dict(fields)
    elif st:
        base, over = st[-1][0], dict(st[-1][1], **fields)
    else:
        base, over = default(), dict(fields)
    st = _push(base, over)
    try:
        yield st[-1][2]
    finally:
        st.pop()


@contextlib.contextmanager
def solve_scope(cfg, pinned=()):
    """the configuration a solve (solver.AdmmRun) was constructed with, for one of its calls: replaces whatever full configuration
    is in force and keeps the field overrides of enclosing use(**fields) blocks -- except for the fields in `pinned`, which the
    solve's constructor was given explicitly (config= pins all of them, conv_precision= the precision).  SCHEDULING_FIELDS are
    never pinned: they say how the launches are issued, not what is computed (`with config.use(streams=1):` around a step of a
    `config=` run keeps that step on one stream -- solver.PartLanes and bench.single_stream_launch_log rely on it)"""
    st = getattr(_tls, 'stack', None)
    over = {k: v for k, v in st[-1][1].items() if k not in pinned or k in SCHEDULING_FIELDS} if st else {}
    st = _push(cfg, over)
    try:
        yield st[-1][2]
    finally:
        st.pop()


FIELDS = tuple(f.name for f in dataclasses.fields(Config))
SCHEDULING_FIELDS = ('streams',)


# the value of scipnp_twostage_ffdnet_args.conv_form (include/scipnp.h) that names a configuration's arithmetic
CONV_FORM_SPLIT_F16, CONV_FORM_F32_WINO_F2, CONV_FORM_F32_WINO_F4 = 1, 2, 3


def conv_form(cfg=None):
    cfg = cfg or current()
    if cfg.precision == 'f16x3':
        return CONV_FORM_SPLIT_F16
    return CONV_FORM_F32_WINO_F4 if cfg.wino_f4 else CONV_FORM_F32_WINO_F2
