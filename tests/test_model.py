"""Model compiler known-answer tests (total mass, per-geom masses,
hinge axes) and a cross-check of model_spec against the reference XML when it
is mounted (never on the GPU box)."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from jitterbug_amd import model, model_spec

REF_XML = "/root/reference/jitterbug_dmc/jitterbug.xml"


def test_layout_matches_header():
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "jitterbug_model.h")).read()
    assert "/* 612 */" in hdr
    assert model.NPARAM == 612
    assert model.P_HINGE == 24 + 10 * 12 and model.P_GEOM == model.P_HINGE + 9 * 8


def test_geom_and_body_masses(params):
    gm = model.geom_masses()
    # SURVEY.md §8a row 1 hand-derived values
    np.testing.assert_allclose(gm[0], 1.1180e-4, rtol=2e-4)     # coreBody1
    np.testing.assert_allclose(gm[1], 2.3654e-3, rtol=2e-4)     # coreBody2
    np.testing.assert_allclose(gm[2], 1.1611e-3, rtol=2e-4)     # screw1
    np.testing.assert_allclose(gm[3], 1.3547e-3, rtol=2e-4)     # screw2
    np.testing.assert_allclose(gm[4], 4.1622e-4, rtol=2e-4)     # leg2 upper cylinder (L=.04624)
    np.testing.assert_allclose(gm[5], 7.321e-6, rtol=2e-4)      # knee tip sphere
    np.testing.assert_allclose(gm[6], 5.8508e-4, rtol=2e-4)     # lower cylinder (L=.065)
    np.testing.assert_allclose(gm[7], 1.2441e-4, rtol=2e-4)     # foot
    np.testing.assert_allclose(gm[12], 3.9028e-4, rtol=2e-4)    # leg1 upper cylinder (L=.04336)
    np.testing.assert_allclose(gm[20], 3.0238e-4, rtol=2e-4)    # threadMass
    np.testing.assert_allclose(gm[21], 7.7409e-3, rtol=2e-4)    # mass ellipsoid
    masses = np.array([model.body_mass(params, b) for b in range(10)])
    np.testing.assert_allclose(masses[0], 4.9930e-3, rtol=2e-4)
    np.testing.assert_allclose(masses[9], 8.0433e-3, rtol=2e-4)
    np.testing.assert_allclose(masses[1:9].sum(), 4.4802e-3, rtol=2e-4)
    np.testing.assert_allclose(masses.sum(), 1.75165e-2, rtol=1e-5)


def test_hinge_axes(params):
    exp = {0: (-.53330, -.84593, 0), 2: (.53330, -.84593, 0), 4: (.65450, -.75606, 0), 6: (-.65450, -.75606, 0), 8: (0, 0, 1)}
    for h, ax in exp.items():
        o = model.P_HINGE + h * model.HINGE_STRIDE + model.H_AXIS
        np.testing.assert_allclose(params[o:o + 3], ax, atol=1e-5)
        if h < 8:   # knee axis equals shoulder axis in the nominal model
            o2 = model.P_HINGE + (h + 1) * model.HINGE_STRIDE + model.H_AXIS
            np.testing.assert_allclose(params[o2:o2 + 3], ax, atol=1e-5)


def test_inertia_positive_definite_and_invweight(params):
    for b in range(10):
        o = model.P_BODY + b * model.BODY_STRIDE
        ii = params[o + model.B_INERTIA:o + model.B_INERTIA + 6]
        I = np.array([[ii[0], ii[3], ii[4]], [ii[3], ii[1], ii[5]], [ii[4], ii[5], ii[2]]])
        w = np.linalg.eigvalsh(I)
        assert w.min() > 0
        assert w[0] + w[1] >= w[2] * (1 - 1e-9)          # triangle inequality of a physical inertia
        assert params[o + model.B_INVW_TRAN] > 0 and params[o + model.B_INVW_ROT] > 0
    M, _, _ = model.mass_matrix_qpos0(params)
    assert np.linalg.eigvalsh(M).min() > 0
    # translational block of M is total mass x identity
    np.testing.assert_allclose(M[:3, :3], np.eye(3) * 1.75165e-2, rtol=1e-5, atol=1e-12)


def test_actuator_law(params):
    gear = params[model.P_GEAR]
    b2 = params[model.P_BIASPRM + 2]
    # tau = gear*clip(u) + gear^2*b2*qd  ->  0.00833 u - 5.5511e-5 qd ; free-run 150.06 rad/s
    np.testing.assert_allclose(-gear * gear * b2, 5.5511e-5, rtol=1e-4)
    np.testing.assert_allclose(gear / (-gear * gear * b2), 150.06, rtol=1e-4)


def _floats(s):
    return [float(x) for x in s.split()]


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference not mounted")
def test_spec_matches_reference_xml():
    """model_spec.SPEC restates jitterbug.xml number for number."""
    root = ET.parse(REF_XML).getroot()
    assert root.find("compiler").attrib["coordinate"] == "global"
    assert float(root.find("option").attrib["timestep"]) == model_spec.SPEC["timestep"]
    assert float(root.find("default/geom").attrib["density"]) == model_spec.DEFAULT_DENSITY
    jb = [b for b in root.find("worldbody").findall("body") if b.attrib["name"] == "jitterbug"][0]
    assert tuple(_floats(jb.attrib["pos"])) == model_spec.SPEC["root"]["pos"]

    def check_geoms(xml_geoms, spec_geoms):
        assert len(xml_geoms) == len(spec_geoms)
        for xg, sg in zip(xml_geoms, spec_geoms):
            assert xg.attrib["type"] == sg["type"]
            assert tuple(_floats(xg.attrib["size"])) == tuple(sg["size"])
            for key in ("pos", "fromto"):
                if key in xg.attrib:
                    assert tuple(_floats(xg.attrib[key])) == tuple(sg[key]), (xg.attrib, sg)
            if "density" in xg.attrib:
                assert float(xg.attrib["density"]) == sg["density"]
            else:
                assert "density" not in sg

    def check_joint(xj, sj):
        assert tuple(_floats(xj.attrib["pos"])) == tuple(sj["pos"])
        assert tuple(_floats(xj.attrib["axis"])) == tuple(sj["axis"])
        assert float(xj.attrib.get("stiffness", 0)) == sj["stiffness"]
        assert float(xj.attrib.get("damping", 0)) == sj["damping"]

    check_geoms(jb.findall("geom"), model_spec.SPEC["root"]["geoms"])
    kids = jb.findall("body")
    assert [k.attrib["name"] for k in kids] == ["leg2upper", "leg3upper", "leg1upper", "leg4upper", "mass"]
    for k, leg in zip(kids[:4], model_spec.SPEC["legs"]):
        check_geoms(k.findall("geom"), leg["upper"]["geoms"])
        check_joint(k.find("joint"), leg["upper"]["joint"])
        lo = k.find("body")
        check_geoms(lo.findall("geom"), leg["lower"]["geoms"])
        check_joint(lo.find("joint"), leg["lower"]["joint"])
    check_geoms(kids[4].findall("geom"), model_spec.SPEC["mass"]["geoms"])
    check_joint(kids[4].find("joint"), model_spec.SPEC["mass"]["joint"])
    act = root.find("actuator/general").attrib
    a = model_spec.SPEC["actuator"]
    assert float(act["gear"]) == a["gear"]
    assert tuple(_floats(act["ctrlrange"])) == a["ctrlrange"]
    assert tuple(_floats(act["gainprm"])) == a["gainprm"]
    assert tuple(_floats(act["biasprm"])) == a["biasprm"]
    assert act["biastype"] == "affine" and act["gaintype"] == "fixed" and act["dyntype"] == "none"
    tgt = [b for b in root.find("worldbody").findall("body") if b.attrib["name"] == "target"][0]
    assert _floats(tgt.find("geom").attrib["pos"])[2] == model_spec.SPEC["target"]["pos"][2]
