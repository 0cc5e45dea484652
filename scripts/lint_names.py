#!/usr/bin/env python3
"""development: undefined-name check of the package's Python sources without third-party tools (pyflakes is not in the image).

For every module: names bound at module level (assignments, defs, classes, imports, ``global`` targets), and for every function the names
it binds itself (arguments, assignments, comprehension targets, ``nonlocal`` / enclosing scopes); every ``Name`` that is loaded and bound
nowhere on that chain and is not a builtin is reported.  Conservative (``from x import *``, ``exec`` are not followed); exits 1 on findings.
usage: lint_names.py [files or directories ...]   (default: eas_snn_amd, bench.py, __graft_entry__.py, tests, scripts, oracle)"""
import ast
import builtins
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILTINS = set(dir(builtins)) | {'__file__', '__name__', '__doc__', '__path__', '__spec__', '__builtins__', '__class__'}


class Scope:
    def __init__(self, node, parent):
        self.node, self.parent, self.bound, self.globals = node, parent, set(), set()


def bind_targets(t, out):
    if isinstance(t, ast.Name):
        out.add(t.id)
    elif isinstance(t, (ast.Tuple, ast.List)):
        for e in t.elts:
            bind_targets(e, out)
    elif isinstance(t, ast.Starred):
        bind_targets(t.value, out)


def collect_bound(body_nodes, scope):
    """names bound directly in this scope (not inside nested function / class scopes, except their own names)"""
    stack = list(body_nodes)
    while stack:
        n = stack.pop()
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            scope.bound.add(n.name)
            stack.extend(n.decorator_list)
            continue
        if isinstance(n, ast.Lambda):
            continue
        if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
            continue            # own scope (handled in check)
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                scope.bound.add((a.asname or a.name).split('.')[0])
        elif isinstance(n, (ast.Assign,)):
            for t in n.targets:
                bind_targets(t, scope.bound)
        elif isinstance(n, (ast.AugAssign, ast.AnnAssign)):
            bind_targets(n.target, scope.bound)
        elif isinstance(n, (ast.For, ast.AsyncFor)):
            bind_targets(n.target, scope.bound)
        elif isinstance(n, (ast.With, ast.AsyncWith)):
            for it in n.items:
                if it.optional_vars is not None:
                    bind_targets(it.optional_vars, scope.bound)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            scope.bound.add(n.name)
        elif isinstance(n, ast.Global):
            scope.globals.update(n.names)
        elif isinstance(n, ast.Nonlocal):
            scope.bound.update(n.names)
        elif isinstance(n, ast.NamedExpr):
            bind_targets(n.target, scope.bound)
        elif hasattr(ast, 'Match') and isinstance(n, ast.Match):
            pass
        stack.extend(ast.iter_child_nodes(n))


def check(path):
    src = open(path).read()
    try:
        tree = ast.parse(src, path)
    except SyntaxError as e:
        return [f'{path}:{e.lineno}: syntax error: {e.msg}']
    problems = []
    mod = Scope(tree, None)
    collect_bound(tree.body, mod)
    # names assigned through ``global`` inside functions are module names too
    for n in ast.walk(tree):
        if isinstance(n, ast.Global):
            mod.bound.update(n.names)
    star = any(isinstance(n, ast.ImportFrom) and any(a.name == '*' for a in n.names) for n in ast.walk(tree))

    def visit(node, scope, in_class=False):
        for child in ast.iter_child_nodes(node):
            if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
                s = Scope(child, scope)
                a = child.args
                for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                    s.bound.add(arg.arg)
                for d in a.defaults + [d for d in a.kw_defaults if d is not None]:
                    visit_expr(d, scope)
                if not isinstance(child, ast.Lambda):
                    for d in child.decorator_list:
                        visit_expr(d, scope)
                    collect_bound(child.body, s)
                    visit(ast.Module(body=child.body, type_ignores=[]), s)
                else:
                    visit_expr(child.body, s)
            elif isinstance(child, ast.ClassDef):
                s = Scope(child, scope)
                collect_bound(child.body, s)
                for b in child.bases + child.decorator_list + [k.value for k in child.keywords]:
                    visit_expr(b, scope)
                # class-level names are visible in the class body only, not in its methods
                visit_class(child, s, scope)
            elif isinstance(child, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
                visit_expr(child, scope)
            else:
                if isinstance(child, ast.Name) and isinstance(child.ctx, ast.Load):
                    use(child, scope)
                visit(child, scope)

    def visit_class(cls, cls_scope, outer):
        for stmt in cls.body:
            if isinstance(stmt, (ast.FunctionDef, ast.AsyncFunctionDef)):
                visit(ast.Module(body=[stmt], type_ignores=[]), _method_parent(cls_scope, outer))
            else:
                visit(ast.Module(body=[stmt], type_ignores=[]), cls_scope)

    def _method_parent(cls_scope, outer):
        # methods resolve free names in the enclosing (function / module) scope, skipping the class scope; defaults and decorators see
        # the class scope -- approximated by a scope that binds nothing new on top of ``outer`` but tolerates class-level names
        s = Scope(cls_scope.node, outer)
        s.bound = set(cls_scope.bound)     # tolerant: a decorator / default may use a class-level name
        return s

    def visit_expr(e, scope):
        if isinstance(e, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
            s = Scope(e, scope)
            for g in e.generators:
                bind_targets(g.target, s.bound)
            for g in e.generators:
                visit_expr(g.iter, s)
                for c in g.ifs:
                    visit_expr(c, s)
            for part in ([e.key, e.value] if isinstance(e, ast.DictComp) else [e.elt]):
                visit_expr(part, s)
            return
        if isinstance(e, ast.Lambda):
            visit(ast.Module(body=[ast.Expr(e)], type_ignores=[]), scope)
            return
        if isinstance(e, ast.Name):
            if isinstance(e.ctx, ast.Load):
                use(e, scope)
            return
        for c in ast.iter_child_nodes(e):
            if isinstance(c, ast.expr) or isinstance(c, (ast.comprehension, ast.keyword, ast.arguments, ast.FormattedValue)):
                visit_expr(c, scope) if isinstance(c, ast.expr) else [visit_expr(x, scope) for x in ast.iter_child_nodes(c) if isinstance(x, ast.expr)]

    def use(name, scope):
        s = scope
        while s is not None:
            if name.id in s.bound or name.id in s.globals:
                return
            s = s.parent
        if name.id in BUILTINS or star:
            return
        problems.append(f'{os.path.relpath(path, ROOT)}:{name.lineno}: undefined name {name.id!r}')

    visit(tree, mod)
    return sorted(set(problems))


def main():
    targets = sys.argv[1:] or [os.path.join(ROOT, p) for p in ('eas_snn_amd', 'bench.py', '__graft_entry__.py', 'tests', 'scripts', 'oracle')]
    files = []
    for t in targets:
        if os.path.isdir(t):
            for d, _, fs in os.walk(t):
                files += [os.path.join(d, f) for f in fs if f.endswith('.py')]
        elif t.endswith('.py'):
            files.append(t)
    bad = []
    for f in sorted(files):
        bad += check(f)
    print('\n'.join(bad) if bad else f'{len(files)} files: no undefined names')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
