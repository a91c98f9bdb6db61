"""The wire types of es_distributed (es.py:12-23 Config / Task / Result, ga.py:4 GATask) under the reference's own
module names, so that what a HIP worker pickles onto Redis is what a reference master unpickles, and the reverse
(dist.py:19-24 pickles (task_id, Result) and Task with protocol -1; pickle stores a namedtuple as
"module.qualname" + fields, so the module path is part of the wire format).

  * the host application already imported es_distributed.es (the drivers are routed from the
    reference's own main.py): its classes are reused -- one class object per process, nothing to translate;
  * otherwise the classes are defined here with __module__ = 'es_distributed.es' / 'es_distributed.ga' and tiny
    stand-in modules holding just these names are registered under sys.modules, so both pickling (which checks that
    the named module really holds the class) and unpickling of reference-made payloads resolve.  A real
    es_distributed package that is importable is imported in preference to the stand-ins; one that exists on the
    path but cannot be imported (its TensorFlow / redis dependencies are absent) is shadowed.
"""
import importlib
import importlib.util
import sys
import types
from collections import namedtuple

_FIELDS = {
    ('es_distributed.es', 'Config'): ['l2coeff', 'noise_stdev', 'episodes_per_batch', 'timesteps_per_batch',
                                      'calc_obstat_prob', 'eval_prob', 'snapshot_freq',
                                      'return_proc_mode', 'episode_cutoff_mode'],
    ('es_distributed.es', 'Task'): ['params', 'ob_mean', 'ob_std', 'ref_batch', 'timestep_limit'],
    ('es_distributed.es', 'Result'): ['worker_id', 'noise_inds_n', 'returns_n2', 'signreturns_n2', 'lengths_n2',
                                      'eval_return', 'eval_length', 'ob_sum', 'ob_sumsq', 'ob_count'],
    ('es_distributed.ga', 'GATask'): ['params', 'population', 'ob_mean', 'ob_std', 'timestep_limit'],
    # es_modified.py:18-23: the VINE variant's Result carries the behaviour characterisations
    ('es_distributed.es_modified', 'Result'): ['worker_id', 'noise_inds_n', 'returns_n2', 'signreturns_n2', 'lengths_n2',
                                               'eval_return', 'eval_length', 'ob_sum', 'ob_sumsq', 'ob_count', 'bc_vectors'],
}


def _reference_module(name):
    """the real reference module if this process has it (or can have it), else None"""
    if name in sys.modules and not getattr(sys.modules[name], '__dne_standin__', False):
        return sys.modules[name]
    if 'es_distributed' in sys.modules and getattr(sys.modules['es_distributed'], '__dne_standin__', False):
        return None
    before = set(sys.modules)
    try:
        if 'es_distributed' not in sys.modules and importlib.util.find_spec('es_distributed') is None:
            return None
        return importlib.import_module(name)
    except Exception:            # present on the path but not importable here (TensorFlow, redis, gym ... missing)
        for k in [k for k in sys.modules if k not in before and (k == 'es_distributed' or k.startswith('es_distributed.'))]:
            del sys.modules[k]   # only what this attempt half-imported; the host application's own modules stay
        return None


def _standin(name):
    """the module `name` to hang a wire type on: a real (already imported) one is used as it is"""
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__dne_standin__ = True
        m.__doc__ = 'stand-in registered by dne_hip.compat: holds only the pickled wire types of ' + name
        sys.modules[name] = m
        if '.' in name:
            parent, child = name.rsplit('.', 1)
            setattr(_standin(parent), child, m)
        else:
            m.__path__ = []
    return sys.modules[name]


def _wire_type(module, name):
    ref = _reference_module(module)
    if ref is not None and hasattr(ref, name):
        cls = getattr(ref, name)
        assert list(cls._fields) == _FIELDS[(module, name)], (module, name, cls._fields)
        return cls
    cls = namedtuple(name, _FIELDS[(module, name)])
    cls.__module__ = module
    setattr(_standin(module), name, cls)
    return cls


Config = _wire_type('es_distributed.es', 'Config')
Task = _wire_type('es_distributed.es', 'Task')
Result = _wire_type('es_distributed.es', 'Result')
GATask = _wire_type('es_distributed.ga', 'GATask')
ModifiedResult = _wire_type('es_distributed.es_modified', 'Result')
