"""The Jitterbug ("KatitaV5") rigid-body model as plain data.

Every number below restates an attribute of the reference model file
(reference: jitterbug_dmc/jitterbug.xml, line cited per entry).  The reference
file is MJCF with ``<compiler coordinate="global"/>`` (jitterbug.xml:4), so all
positions / fromto / axes are WORLD coordinates at the reference configuration.

The layout (bodies -> joint + geoms) is this package's own; ``model.compile``
turns it into the flat parameter table of ``include/jitterbug_model.h``.
``tests/test_model.py`` re-parses the reference XML (when /root/reference is
mounted) and checks that this table matches it number for number.
"""
import copy

# geom density default (jitterbug.xml:22)
DEFAULT_DENSITY = 7700.0

LEG_NAMES = ("leg2", "leg3", "leg1", "leg4")      # XML depth-first order


def _leg(name, up_fromto, tip_pos, j1_pos, j1_axis, lo_fromto, foot_pos, j2_pos, j2_axis):
    return dict(
        name=name,
        upper=dict(
            geoms=[
                dict(type="cylinder", fromto=up_fromto, size=(0.00061,)),
                dict(type="sphere", pos=tip_pos, size=(0.00061,)),
            ],
            joint=dict(pos=j1_pos, axis=j1_axis, stiffness=0.4, damping=0.00001),
        ),
        lower=dict(
            geoms=[
                dict(type="cylinder", fromto=lo_fromto, size=(0.00061,)),
                dict(type="sphere", pos=foot_pos, size=(0.003,), density=1100.0),
            ],
            joint=dict(pos=j2_pos, axis=j2_axis, stiffness=0.4, damping=0.00001),
        ),
    )


SPEC = dict(
    timestep=0.0002,                          # jitterbug.xml:18
    gravity=(0.0, 0.0, -9.81),                # MuJoCo default (not set in the file)
    solref=(0.02, 1.0),                       # MuJoCo default
    solimp=(0.9, 0.95, 0.001, 0.5, 2.0),      # MuJoCo 2.0 default
    friction=1.0,                             # MuJoCo default geom friction[0]
    impratio=1.0,
    root=dict(
        pos=(0.0, 0.0, 0.035),                # jitterbug.xml:37
        geoms=[                               # jitterbug.xml:44-47
            dict(name="coreBody1", type="box", size=(0.0055, 0.006, 0.011), pos=(0.0, 0.0, 0.04), density=38.5),
            dict(name="coreBody2", type="box", size=(0.008, 0.006, 0.008), pos=(0.0, 0.0, 0.021), density=770.0),
            dict(name="screw1", type="cylinder", size=(0.002,), fromto=(0.0, 0.0, 0.026, 0.0, 0.012, 0.026)),
            dict(name="screw2", type="ellipsoid", size=(0.014, 0.006, 0.0005), pos=(0.0, 0.017, 0.026)),
        ],
    ),
    legs=[
        # leg 2, jitterbug.xml:52-63
        _leg("leg2", (0.0287, -0.0308, 0.07, 0.005, 0.0035, 0.05), (0.0287, -0.0308, 0.07),
             (0.003, 0.0035, 0.049), (-2.32, -3.68, 0.0),
             (0.0287, -0.0308, 0.005, 0.0287, -0.0308, 0.07), (0.0287, -0.0308, 0.005),
             (0.0287, -0.0308, 0.068), (-2.32, -3.68, 0.0)),
        # leg 3, jitterbug.xml:66-76
        _leg("leg3", (-0.0287, -0.0308, 0.07, -0.005, 0.0035, 0.05), (-0.0287, -0.0308, 0.07),
             (-0.003, 0.0035, 0.049), (2.32, -3.68, 0.0),
             (-0.0287, -0.0308, 0.005, -0.0287, -0.0308, 0.07), (-0.0287, -0.0308, 0.005),
             (-0.0287, -0.0308, 0.068), (2.32, -3.68, 0.0)),
        # leg 1, jitterbug.xml:80-90
        _leg("leg1", (0.0287, 0.0328, 0.068, 0.003, 0.0035, 0.049), (0.0287, 0.0328, 0.068),
             (0.003, 0.0035, 0.049), (2.32, -2.68, 0.0),
             (0.0287, 0.0328, 0.003, 0.0287, 0.0328, 0.068), (0.0287, 0.0328, 0.003),
             (0.0287, 0.0328, 0.068), (2.32, -2.68, 0.0)),
        # leg 4, jitterbug.xml:93-103
        _leg("leg4", (-0.0287, 0.0328, 0.068, -0.003, 0.0035, 0.049), (-0.0287, 0.0328, 0.068),
             (-0.003, 0.0035, 0.049), (-2.32, -2.68, 0.0),
             (-0.0287, 0.0328, 0.003, -0.0287, 0.0328, 0.068), (-0.0287, 0.0328, 0.003),
             (-0.0287, 0.0328, 0.068), (-2.32, -2.68, 0.0)),
    ],
    mass=dict(                                # jitterbug.xml:105-109
        geoms=[
            dict(name="threadMass", type="cylinder", size=(0.001,), fromto=(0.0, -0.004, 0.05, 0.0, -0.004, 0.0625)),
            dict(name="mass", type="ellipsoid", size=(0.008, 0.01, 0.003), pos=(0.0043, -0.004, 0.061)),
        ],
        joint=dict(pos=(0.0, -0.004, 0.05), axis=(0.0, 0.0, 1.0), stiffness=0.0, damping=0.0),
    ),
    target=dict(pos=(0.0, 0.0, 0.035)),       # jitterbug.xml:114-115 (geom world z)
    actuator=dict(                            # jitterbug.xml:129-145
        ctrlrange=(-1.0, 1.0), gear=0.00833, gainprm=(1.0, 0.0, 0.0), biasprm=(0.0, 0.0, -0.8),
    ),
)


def default_spec():
    """A deep copy of the nominal model spec, safe to perturb."""
    return copy.deepcopy(SPEC)
