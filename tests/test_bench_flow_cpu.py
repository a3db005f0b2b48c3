"""Static guard on bench.py's multi-rank control flow: a training step contains collectives (ITC all-gather, gradient
all-reduce), so nothing that runs a step or a collective may sit under `if rank == 0` - rank 0 would wait for peers that
never come (this deadlocked every N > 1 run once)."""
import ast
import os

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
FORBIDDEN = {"step", "_step", "eager_step", "fwd_bwd", "runner", "timed_gemms", "model", "fence"}


def _is_rank0_test(node):
    return (isinstance(node, ast.Compare) and isinstance(node.left, ast.Name) and node.left.id == "rank" and
            len(node.ops) == 1 and isinstance(node.ops[0], ast.Eq) and isinstance(node.comparators[0], ast.Constant) and
            node.comparators[0].value == 0)


def _calls(nodes):
    for n in nodes:
        for c in ast.walk(n):
            if isinstance(c, ast.Call):
                f = c.func
                if isinstance(f, ast.Name):
                    yield f.id, c.lineno
                elif isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name):
                    yield "%s.%s" % (f.value.id, f.attr), c.lineno


def test_no_step_or_collective_under_rank0_only():
    tree = ast.parse(open(SRC).read())
    bad = []
    for node in ast.walk(tree):
        if isinstance(node, ast.If) and _is_rank0_test(node.test):
            for name, line in _calls(node.body):
                if name in FORBIDDEN or name.startswith("dist."):
                    bad.append((name, line))
    assert not bad, "rank-0-only code runs steps / collectives: %s" % bad


def test_contract_fields_present():
    src = open(SRC).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"workload"', '"bound"', '"achieved"',
                '"peak"', '"frac"', '"traffic"', '"cores"', '"kind"', '"sample"'):
        assert key in src, key
