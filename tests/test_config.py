"""CPU: adaptivepnp_sci_amd.config -- the ONE configuration object behind every choice of kernel form (precision, fp32 form,
F(4x4) on / off, weight-gradient form, side streams, ADMM-TV paths); the environment variables are its process default only."""
import threading

import pytest


def test_defaults_environment_and_validation(monkeypatch):
    from adaptivepnp_sci_amd import config, nets, ops
    for k in config._ENV:
        monkeypatch.delenv(k, raising=False)
    c = config.current()
    assert (c.precision, c.f32_form, c.wino_f4, c.f32_wgrad, c.streams, c.hipgraph, c.tv_defer, c.wgrad_slabs) == \
        ('f32', 'winograd', True, 'f4', 2, False, True, None)
    assert nets.default_precision() == 'f32' and ops.wino_f4_enabled() and ops.side_stream_count() == 2
    # the environment is the process default, consulted through current() only, and a change is seen
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', 'f16x3')
    monkeypatch.setenv('SCIPNP_WINO_F4', '0')
    monkeypatch.setenv('SCIPNP_STREAMS', '1')
    assert nets.default_precision() == 'f16x3' and not ops.wino_f4_enabled() and ops.side_stream_count() == 1
    monkeypatch.setenv('SCIPNP_FFDNET_PRECISION', 'f32')            # the alias loses against the main variable
    assert config.current().precision == 'f16x3'
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', 'bf16')
    with pytest.raises(ValueError, match='SCIPNP_CONV_PRECISION'):
        config.current()
    monkeypatch.delenv('SCIPNP_CONV_PRECISION')
    for bad in (dict(streams=0), dict(streams=9), dict(f32_form='im2col'), dict(f32_wgrad='f8'), dict(wgrad_slabs=0)):
        with pytest.raises(ValueError):
            config.Config(**bad)
    with pytest.raises(Exception):
        config.current().precision = 'f16x3'                          # frozen: a configuration is a value


def test_explicit_configurations_win_over_the_environment_and_nest(monkeypatch):
    from adaptivepnp_sci_amd import config, finetune, nets, ops
    monkeypatch.setenv('SCIPNP_CONV_PRECISION', 'f16x3')
    with config.use(config.Config()):                                # a full configuration: nothing of the environment is left
        assert nets.default_precision() == 'f32'
        with config.use(wino_f4=False, f32_wgrad='f2'):
            assert not ops.wino_f4_enabled() and not finetune.wgrad_f4_enabled() and nets.default_precision() == 'f32'
        assert ops.wino_f4_enabled() and finetune.wgrad_f4_enabled()
    assert nets.default_precision() == 'f16x3'
    config.set_default(config.Config(streams=3))
    try:
        assert config.current().streams == 3 and config.current().precision == 'f32'     # explicit default: env ignored
    finally:
        config.set_default(None)
    assert config.current().precision == 'f16x3'


def test_a_solves_configuration_field_overrides_and_pins():
    """solve_scope = what AdmmRun.step does: the run's own configuration, outer FIELD overrides still reaching in -- unless the
    run's constructor pinned that field (conv_precision= pins the precision, config= everything)"""
    from adaptivepnp_sci_amd import config
    run_cfg = config.Config(precision='f16x3', streams=2)
    with config.use(streams=1, precision='f32'):
        with config.solve_scope(run_cfg):
            assert (config.current().precision, config.current().streams) == ('f32', 1)
        with config.solve_scope(run_cfg, pinned=('precision',)):
            assert (config.current().precision, config.current().streams) == ('f16x3', 1)
        with config.solve_scope(run_cfg, pinned=config.FIELDS):
            # config= pins every field that says WHAT is computed; the scheduling fields (streams) stay overridable: PartLanes and
            # bench.single_stream_launch_log wrap steps of such runs in use(streams=1)
            assert config.current() == run_cfg.replace(streams=1)
            assert config.current().precision == 'f16x3'
        with config.use(config.Config(hipgraph=True)):                # a full configuration resets the field overrides
            assert config.current().streams == 2 and config.current().hipgraph
    assert config.conv_form(config.Config(precision='f16x3')) == 1 and config.conv_form(config.Config(wino_f4=False)) == 2 \
        and config.conv_form(config.Config()) == 3


def test_configurations_are_per_thread():
    from adaptivepnp_sci_amd import config
    seen = {}

    def worker():
        seen['inner'] = config.current().streams
        with config.use(streams=5):
            seen['own'] = config.current().streams

    with config.use(streams=4):
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert config.current().streams == 4
    assert seen['inner'] == config.default().streams and seen['own'] == 5
