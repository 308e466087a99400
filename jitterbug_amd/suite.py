"""``suite.load`` for the Jitterbug domains (dm_control.suite.load signature; reference README.md:54-58,
benchmarks/benchmark.py:138-143).  When dm_control is importable the domains are ALSO registered into
``dm_control.suite`` the way the reference does (reference jitterbug_dmc/__init__.py:31-43)."""
from . import augmented_jitterbug, jitterbug

_DOMAINS = {"jitterbug": jitterbug, "augmented_jitterbug": augmented_jitterbug}


def load(domain_name, task_name, task_kwargs=None, environment_kwargs=None, visualize_reward=False):
    if domain_name not in _DOMAINS:
        raise ValueError("Domain {!r} does not exist.".format(domain_name))
    domain = _DOMAINS[domain_name]
    if task_name not in domain.SUITE:
        raise ValueError("Level {!r} does not exist in domain {!r}.".format(task_name, domain_name))
    task_kwargs = dict(task_kwargs or {})
    if environment_kwargs is not None:
        task_kwargs["environment_kwargs"] = dict(environment_kwargs)
    return domain.SUITE[task_name](**task_kwargs)       # visualize_reward only recolours geoms in the reference: no-op here


def register_with_dm_control():
    """Best effort: add the domains to dm_control.suite if it is installed (it is not in this image)."""
    try:
        from dm_control import suite as dmc_suite
    except Exception:
        return False
    dmc_suite._DOMAINS.update(_DOMAINS)
    dmc_suite.ALL_TASKS = dmc_suite._get_tasks(tag=None)
    dmc_suite.BENCHMARKING = dmc_suite._get_tasks("benchmarking")
    dmc_suite.EASY = dmc_suite._get_tasks("easy")
    dmc_suite.HARD = dmc_suite._get_tasks("hard")
    dmc_suite.EXTRA = tuple(sorted(set(dmc_suite.ALL_TASKS) - set(dmc_suite.BENCHMARKING)))
    dmc_suite.TASKS_BY_DOMAIN = dmc_suite._get_tasks_by_domain(dmc_suite.ALL_TASKS)
    return True
