// jb_model_compile_types.h — plain data types of the native model compiler / randomiser (jb_model_compile.hpp).
#pragma once
#include "../../include/jitterbug_model.h"

typedef struct JbGeomSpec {
    int type, body, has_fromto;
    double fromto[6], pos[3], size[3];
    double density;                 // < 0: the model's default density
} JbGeomSpec;
typedef struct JbHingeSpec { double pos[3], axis[3], stiffness, damping; } JbHingeSpec;
// the uncompiled model: what jitterbug_amd/model_spec.py holds (world coordinates at qpos0, like the reference's MJCF)
typedef struct JbNominalSpec {
    double timestep, gravity[3], solref[2], solimp[5], friction, impratio;
    double gear, gain, biasprm[3], ctrlrange[2], root_pos[3], target_z, default_density;
    JbGeomSpec geoms[JB_NGEOM];
    JbHingeSpec hinges[JB_NHINGE];
} JbNominalSpec;
