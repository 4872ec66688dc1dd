// <mujoco/mujoco.h> as this build sees it: the slice of MuJoCo's public API the MJPC host code touches (mujoco_min.h).
// Every mjpc/ header includes <mujoco/mujoco.h>, exactly as the reference's do; the Makefile puts this directory on the include
// path. Building against a real MuJoCo is a matter of pointing -I at its headers instead (the struct members and function names
// used are MuJoCo's own) and dropping model_io's blob reader for mj_loadXML.
#pragma once
#include "../../mujoco_min.h"
