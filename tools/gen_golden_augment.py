"""Generates tests/golden/augment_golden.json by RUNNING the reference's augment_Jitterbug (numpy/ElementTree only).

The reference module imports dm_control at the top, which is not installed, so its six pure helper functions are
extracted from the source by AST and executed with numpy / ElementTree / os in scope; `__file__` points
at a scratch directory that holds a copy of the reference jitterbug.xml, where the function writes its output XML.
Only numbers (inputs = seed + flags, outputs = perturbed attributes) are stored. Run from the repo root:
    python tools/gen_golden_augment.py
"""
import ast
import json
import os
import shutil
import tempfile
import xml.etree.ElementTree as ET

import numpy as np

REF = "/root/reference/jitterbug_dmc"
WANT = {"str2array", "array2str", "update_features", "fromto2vect", "augment_Jitterbug", "print_changes"}


def load_functions(workdir):
    src = open(os.path.join(REF, "augmented_jitterbug.py")).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    mod = ast.Module(body=body, type_ignores=[])
    ns = {"np": np, "ET": ET, "os": os, "__file__": os.path.join(workdir, "augmented_jitterbug.py"), "print": lambda *a, **k: None}
    exec(compile(mod, "augmented_jitterbug_extract", "exec"), ns)
    return ns


def floats(s):
    return [float(x) for x in s.split()]


def read_xml(path):
    root = ET.parse(path).getroot()
    jb = [b for b in root.find("worldbody").findall("body") if b.attrib["name"] == "jitterbug"][0]
    out = {"legs": []}
    kids = jb.findall("body")
    for k in kids[:4]:
        lo = k.find("body")
        cyl_u = [g for g in k.findall("geom") if g.attrib["type"] == "cylinder"][0]
        sph_u = [g for g in k.findall("geom") if g.attrib["type"] == "sphere"][0]
        cyl_l = [g for g in lo.findall("geom") if g.attrib["type"] == "cylinder"][0]
        sph_l = [g for g in lo.findall("geom") if g.attrib["type"] == "sphere"][0]
        out["legs"].append(dict(
            name=k.attrib["name"], upper_fromto=floats(cyl_u.attrib["fromto"]), tip_pos=floats(sph_u.attrib["pos"]),
            j1_pos=floats(k.find("joint").attrib["pos"]), j1_axis=floats(k.find("joint").attrib["axis"]),
            lower_fromto=floats(cyl_l.attrib["fromto"]), foot_pos=floats(sph_l.attrib["pos"]),
            j2_pos=floats(lo.find("joint").attrib["pos"]), j2_axis=floats(lo.find("joint").attrib["axis"])))
    m = kids[4]
    out["mass"] = dict(thread_fromto=floats([g for g in m.findall("geom") if g.attrib["name"] == "threadMass"][0].attrib["fromto"]),
                       mass_pos=floats([g for g in m.findall("geom") if g.attrib["name"] == "mass"][0].attrib["pos"]),
                       mass_size=floats([g for g in m.findall("geom") if g.attrib["name"] == "mass"][0].attrib["size"]),
                       joint_pos=floats(m.find("joint").attrib["pos"]))
    out["gear"] = float(root.find("actuator/general").attrib["gear"])
    out["default_density"] = float(root.find("default/geom").attrib["density"])
    out["core_density"] = [float(g.attrib["density"]) for g in jb.findall("geom") if "density" in g.attrib]
    return out


def main():
    work = tempfile.mkdtemp()
    shutil.copy(os.path.join(REF, "jitterbug.xml"), os.path.join(work, "jitterbug.xml"))
    ns = load_functions(work)
    cases = []
    for seed, kw in [(0, dict(modify_legs=True, modify_mass=True)), (1, dict(modify_legs=True, modify_mass=True)),
                     (2, dict(modify_legs=True)), (3, dict(modify_mass=True)),
                     (4, dict(modify_legs=True, modify_mass=True, modify_gear=True, modify_global_density=True,
                              modify_coreBody1=True, modify_coreBody2=True))]:
        np.random.seed(seed)
        ns["augment_Jitterbug"](**kw)
        cases.append(dict(seed=seed, kwargs=kw, xml=read_xml(os.path.join(work, "augmented_jitterbug.xml"))))
    shutil.rmtree(work)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "augment_golden.json")
    json.dump(dict(generator="tools/gen_golden_augment.py", source="reference augmented_jitterbug.py:95-267 executed here", cases=cases),
              open(out, "w"), indent=1)
    print("wrote", out, len(cases), "cases")


if __name__ == "__main__":
    main()
