"""Generates tests/golden/presolve_changes.json: the known-answer cases of the reference's presolve tests
as DATA (input problem, expected `Changes` or error), one entry per test function of
/root/reference/src/data/linear_program/general_form/presolve/test/changes.rs.

The reference's tests are Rust source with the data written as literals; this script translates the
literal syntax (R32!(a, b), vec![..; n], Some/None, struct literals, HashMap insert blocks) into Python
values and dumps them.  Run in the build container (the reference is not present on the GPU box):
    python scripts/gen_presolve_fixtures.py
"""
import json
import os
import re
import sys
from fractions import Fraction

SRC = "/root/reference/src/data/linear_program/general_form/presolve/test/changes.rs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "presolve_changes.json")


def balanced(text, start, open_ch="(", close_ch=")"):
    """index just past the bracket matching text[start] (which must be open_ch)"""
    depth = 0
    for k in range(start, len(text)):
        if text[k] == open_ch:
            depth += 1
        elif text[k] == close_ch:
            depth -= 1
            if depth == 0:
                return k + 1
    raise ValueError("unbalanced")


def split_top(text):
    parts, depth, cur = [], 0, []
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    tail = "".join(cur).strip()
    if tail:
        parts.append(tail)
    return parts


def repeat_vecs(text):
    """[X; n] -> ([X] * n), innermost first"""
    while True:
        m = re.search(r";\s*(\d+)\s*\]", text)
        if not m:
            return text
        close = m.end() - 1
        depth, k = 0, close
        while True:
            if text[k] == "]":
                depth += 1
            elif text[k] == "[":
                depth -= 1
                if depth == 0:
                    break
            k -= 1
        inner = text[k + 1:m.start()]
        text = text[:k] + "([" + inner + "] * " + m.group(1) + ")" + text[close + 1:]


def hashmap_blocks(text):
    """{ let mut x = HashMap::new(); x.insert(k, v); ...; x } -> hm([(k, v), ...])"""
    while True:
        m = re.search(r"\{\s*let mut (\w+) = HashMap::new\(\);", text)
        if not m:
            return text
        end = balanced(text, m.start(), "{", "}")
        body = text[m.end():end - 1]
        name = m.group(1)
        items = []
        pos = 0
        while True:
            k = body.find(name + ".insert(", pos)
            if k < 0:
                break
            a = k + len(name) + len(".insert")
            b = balanced(body, a)
            items.append("(" + body[a + 1:b - 1] + ")")
            pos = b
        text = text[:m.start()] + "hm([" + ", ".join(items) + "])" + text[end:]


def struct_literals(text, name, func):
    """Name { a: x, b: y } -> func(a=x, b=y)"""
    while True:
        m = re.search(re.escape(name) + r"\s*\{", text)
        if not m:
            return text
        end = balanced(text, m.end() - 1, "{", "}")
        body = text[m.end():end - 1]
        fields = []
        for part in split_top(body):
            key, value = part.split(":", 1)
            fields.append(key.strip() + "=" + value.strip())
        text = text[:m.start()] + func + "(" + ", ".join(fields) + ")" + text[end:]


def to_python(expr):
    expr = re.sub(r"//[^\n]*", "", expr)
    expr = hashmap_blocks(expr)
    expr = re.sub(r"R32!\(", "Fr(", expr)
    expr = expr.replace("vec![", "[").replace("&[", "[")
    expr = repeat_vecs(expr)
    expr = expr.replace("HashMap::default()", "hm([])")
    expr = re.sub(r"\bSome\(", "(", expr)
    expr = expr.replace("RangedConstraintRelation::Range(", "rng(")
    for rust, py in (("RangedConstraintRelation::Less", '("L",)'), ("RangedConstraintRelation::Greater", '("G",)'),
                     ("RangedConstraintRelation::Equal", '("E",)'), ("VariableType::Continuous", '"C"'),
                     ("VariableType::Integer", '"I"'), ("Objective::Maximize", '"max"'), ("Objective::Minimize", '"min"'),
                     ("BoundDirection::Lower", "0"), ("BoundDirection::Upper", "1"),
                     ("LinearProgramType::Infeasible", '"infeasible"'), ("LinearProgramType::Unbounded", '"unbounded"'),
                     (".to_string()", ""), ("false", "False"), ("true", "True"),
                     ("ColumnMajor::from_test_data(", "matrix("), ("DenseVector::new(", "dense("),
                     ("RemovedVariable::Solved(", "solved(")):
        expr = expr.replace(rust, py)
    expr = struct_literals(expr, "RemovedVariable::FunctionOfOthers", "function")
    expr = struct_literals(expr, "Variable", "variable")
    expr = struct_literals(expr, "Changes", "changes")
    expr = re.sub(r"\bOk\(", "ok(", expr)
    expr = re.sub(r"\bErr\(", "err(", expr)
    return expr


ENV = {
    "Fr": lambda a, b=1: Fraction(a, b),
    "rng": lambda r: ("R", r),
    "hm": lambda items: dict(items),
    "matrix": lambda rows, ncols: {"rows": rows, "ncols": ncols},
    "dense": lambda values, n: values,
    "solved": lambda v: ("solved", v),
    "function": lambda constant, coefficients: ("function", constant, coefficients),
    "variable": lambda **kw: kw,
    "changes": lambda **kw: kw,
    "ok": lambda c: ("ok", c),
    "err": lambda e: ("err", e),
    "None": None,
}


def jsonable(v):
    if isinstance(v, Fraction):
        return [v.numerator, v.denominator]
    if isinstance(v, dict):
        return {(json.dumps(jsonable(k)) if not isinstance(k, str) else k): jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    return v


def main():
    text = open(SRC).read()
    cases = []
    for m in re.finditer(r"#\[test\]\s*fn (\w+)\(\)", text):
        nxt = text.find("#[test]", m.end())
        chunk = text[m.end():nxt if nxt >= 0 else len(text)]
        g = chunk.find("GeneralForm::new(")
        g_end = balanced(chunk, g + len("GeneralForm::new"))
        args = split_top(to_python(chunk[g + len("GeneralForm::new("):g_end - 1]))
        objective, matrix, ctypes, b, variables, _names, fixed_cost = (eval(a, dict(ENV)) for a in args)
        variables = [dict(v) for v in variables]
        for mm in re.finditer(r"initial\.variables\[(\d+)\]\.(\w+) = ([^;]+);", chunk):
            variables[int(mm.group(1))][mm.group(2)] = eval(to_python(mm.group(3)), dict(ENV))
        a = chunk.find("compute_presolve_changes(),")
        rest = chunk[a + len("compute_presolve_changes(),"):]
        k = re.search(r"\b(Ok|Err)\(", rest)
        e_end = balanced(rest, k.end() - 1)
        expected = eval(to_python(rest[k.start():e_end]), dict(ENV))
        cases.append({"name": m.group(1), "objective": objective, "rows": matrix["rows"], "ncols": matrix["ncols"],
                      "constraint_types": ctypes, "b": b, "variables": variables, "fixed_cost": fixed_cost,
                      "expected": expected})
    with open(OUT, "w") as f:
        json.dump(jsonable(cases), f, indent=1)
    print(len(cases), "cases ->", OUT)


if __name__ == "__main__":
    sys.exit(main())
