"""The C++ host layer (mujoco_mpc_amd/host: namespace mjpc mirror of the reference's Planner / Trajectory /
Task / TimeSpline / State / ThreadPool classes over the C ABI). The test programs are ports of the reference's
gtest suites (cited in each .cc); this file builds and runs them."""
import os
import subprocess

import pytest

from mujoco_mpc_amd import mjcf
from mujoco_mpc_amd.build import build_host
from mujoco_mpc_amd.task import load_task

HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mujoco_mpc_amd", "host")


@pytest.fixture(scope="module")
def blobs(tmp_path_factory):
    build_host()
    d = tmp_path_factory.mktemp("blobs")
    for name in ("Cartpole", "Particle", "ParticleCopy", "QuadrupedFlat", "HumanoidTrack"):
        mjcf.save_blob(load_task(name).model, str(d / f"{name}.mjpx"))
    return str(d)


def run(exe, *args, timeout=300):
    out = subprocess.run([os.path.join(HOST, "build", exe), *args], capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


@pytest.mark.parametrize("exe", ["spline_test", "norm_test", "trajectory_test", "threadpool_test", "utilities_test"])
def test_cpu_suites(blobs, exe):
    assert "OK" in run(exe)


def test_task_and_state_suites(blobs):
    assert "OK" in run("task_test", os.path.join(blobs, "Particle.mjpx"), os.path.join(blobs, "Cartpole.mjpx"),
                       os.path.join(blobs, "QuadrupedFlat.mjpx"), os.path.join(blobs, "HumanoidTrack.mjpx"))
    assert "OK" in run("state_test", os.path.join(blobs, "Particle.mjpx"))


def test_shared_classes_under_thread_sanitizer(blobs):
    """`make tsan` (host/Makefile): State -- one writer through Set / SetTime, three readers through CopyTo, no torn snapshot -- and
    ThreadPool -- rounds of Schedule / WaitCount / ResetCount -- built with -fsanitize=thread, with the reference's own state and
    thread-pool tests beside them; a report ends the run (halt_on_error)"""
    r = subprocess.run(["make", "-C", HOST, "-s", "tsan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    for exe in ("concurrency_test", "state_test", "threadpool_test"):
        out = subprocess.run([os.path.join(HOST, "build", "tsan", exe), os.path.join(blobs, "Particle.mjpx")], capture_output=True, text=True,
                             timeout=600, env=env)
        assert out.returncode == 0 and "OK" in out.stdout and "ThreadSanitizer" not in out.stderr, exe + ": " + out.stderr[-2000:]


def test_host_code_names_mujoco_by_its_public_header():
    """every mjpc/ header reaches MuJoCo's types through <mujoco/mujoco.h> (host/include/mujoco/mujoco.h forwards to the subset this
    build carries), so the host layer is source-compatible with a real MuJoCo include path; a TU that only knows the public name
    compiles"""
    for dirpath, _, files in os.walk(os.path.join(HOST, "mjpc")):
        for f in files:
            if f.endswith((".h", ".cc")):
                assert "mujoco_min.h" not in open(os.path.join(dirpath, f)).read(), f
    src = '#include <mujoco/mujoco.h>\n#include "mjpc/planners/planner.h"\n#include "mjpc/trajectory.h"\n' \
          'int main() { mjModel* m = nullptr; mjData* d = nullptr; mjtNum x = mju_max(1.0, 2.0); (void)m; (void)d; return x > 0 ? 0 : 1; }\n'
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(HOST, "include"), "-I", HOST, "-x", "c++", "-"],
                       input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_host_library_links_the_c_abi(blobs):
    out = subprocess.run(["ldd", os.path.join(HOST, "build", "libmjpc_host.so")], capture_output=True, text=True).stdout
    assert "libmjpcx.so" in out


@pytest.mark.gpu
def test_gpu_sampling_planner_cpp(blobs):
    assert "OK" in run("gpu_planner_test", blobs)


@pytest.mark.gpu
def test_reference_behavioural_suites_cpp(blobs):
    """sampling_planner_test.cc, robust_planner_test.cc and ilqg_test.cc with the reference's own settings on the GPU planners"""
    assert "OK" in run("behaviour_test", blobs)


@pytest.mark.gpu
def test_trajectory_rollout_cpp(blobs):
    """mjpc/test/agent/rollout_test.cc through the restored Trajectory::Rollout / RolloutDiscrete / NoisyRollout (host policy,
    device physics) and Planner::data_ / ResizeMjData"""
    assert "OK" in run("rollout_test", blobs)


@pytest.mark.gpu
def test_agent_cpp(blobs):
    assert "OK" in run("agent_test", blobs)


@pytest.mark.gpu
def test_testspeed_app(blobs):
    out = run("testspeed_app", "--task=Cartpole", "--total_time=0.5", "--steps_per_planning_iteration=4",
              f"--model_dir={blobs}", "--candidates=4096")
    assert "Average cost per step" in out


@pytest.mark.gpu
def test_testspeed_app_quadruped_transition_reads_device_kinematics(blobs):
    """QuadrupedFlat::TransitionLocked (quadruped.cc:229-391) runs every simulation step on mjData kinematics fetched through
    mjpcx_kinematics (automatic gait switching is on in the task XML)"""
    out = run("testspeed_app", "--task=QuadrupedFlat", "--total_time=0.2", "--steps_per_planning_iteration=10",
              f"--model_dir={blobs}", "--candidates=64")
    assert "Average cost per step" in out


def test_host_normal_generator_equals_the_oracle_stream(blobs):
    """HostGaussianPair (mjpc/utilities.h: Philox4x32-10 + Box-Muller) draws what the device and the oracle draw"""
    import ctypes as C
    import numpy as np
    from mujoco_mpc_amd.hostplanner import host_gaussian_pair
    from oracle import pyoracle
    L = pyoracle.lib()
    L.ogaussian_pair.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    for seed, cand, pair, it in ((0, 0, 0, 0), (3, 17, 5, 2), (2**40 + 7, 65535, 1023, 99)):
        z = (C.c_double * 2)()
        L.ogaussian_pair(seed, cand, pair, it, z)
        assert np.allclose(host_gaussian_pair(seed, cand, pair, it), [z[0], z[1]], rtol=1e-14, atol=0)
