#!/usr/bin/env python3
"""Undefined-name check (a pyflakes subset; pyflakes is not in the image): every ``Name`` that is read must be bound
somewhere in its function, an enclosing function, the module, or builtins.  Run before spending a GPU call on code
that cannot execute without a GPU.      python tools/lint_names.py cape_amd/*.py bench.py tests/*.py"""
import ast
import builtins
import sys


def bound_names(node):
    """Names bound directly in this scope (not in nested function/class scopes, except their own names)."""
    out = set()

    def visit(n, top=True):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
            if not top:
                return
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)) and top:
            a = n.args
            for x in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                out.add(x.arg)
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                out.add((al.asname or al.name).split('.')[0])
        if isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        if isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
        for c in ast.iter_child_nodes(n):
            if isinstance(c, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                out.add(c.name)
            elif isinstance(c, ast.Lambda):
                continue
            else:
                visit(c, False)

    visit(node)
    return out


def check(path):
    tree = ast.parse(open(path).read(), path)
    problems = []
    base = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}

    def walk(scope, env):
        env = env | bound_names(scope)
        # comprehension targets are bound_names'ed as Store names already (ast treats them as Name Store)
        for n in ast.iter_child_nodes(scope):
            inner(n, env)

    def inner(n, env):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            for d in getattr(n, 'decorator_list', []):
                inner(d, env)
            for d in n.args.defaults + [k for k in n.args.kw_defaults if k is not None]:
                inner(d, env)
            walk(n, env)
            return
        if isinstance(n, ast.ClassDef):
            for b in n.bases + n.decorator_list:
                inner(b, env)
            walk(n, env)
            return
        if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
            env = env | {t.id for g in n.generators for t in ast.walk(g.target) if isinstance(t, ast.Name)}
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in env:
            problems.append("%s:%d: undefined name %r" % (path, n.lineno, n.id))
        for c in ast.iter_child_nodes(n):
            inner(c, env)

    walk(tree, base)
    return problems


if __name__ == "__main__":
    bad = [p for f in sys.argv[1:] for p in check(f)]
    print("\n".join(bad) if bad else "names ok (%d files)" % len(sys.argv[1:]))
    sys.exit(1 if bad else 0)
