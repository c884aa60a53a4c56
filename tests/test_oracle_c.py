"""Pins the C oracle (oracle/ppn_oracle.c) against the numpy/scipy oracle (itself pinned by the reference's
known answers): same scenarios as the engine checks, driven through the shared harness."""
import os
import subprocess

import pytest

import engine_checks as ec
from helpers import ROOT
from test_oracle_known_answers import _basic_topology_policy

LIB = os.path.join(ROOT, 'oracle', '_build', 'liboracle.so')


@pytest.fixture(scope='session')
def orc():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    return LIB


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
@pytest.mark.parametrize('env', ['default14_for_tests', 'default14_for_tests_hard_overflow'])
def test_c_oracle_do_nothing(orc, env, solver):
    ec.check_do_nothing(orc, env, solver)


def test_c_oracle_dc(orc):
    ec.check_do_nothing(orc, 'default14_for_tests_beta', 'fdxb', steps=8, batch=1)


def test_c_oracle_config1_default14_dc_1000_steps(orc):
    ec.check_config1_default14_dc(orc)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_c_oracle_hard_overflow_scenario(orc, solver):
    ec.check_hard_overflow_scenario(orc, solver)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_c_oracle_k3(orc, solver):
    nodes = list(range(1, 15))
    flags = ec.check_topology_scenarios(orc, 'default14_for_tests_alpha', nodes, 7, _basic_topology_policy, solver)
    for node, f in zip(nodes, flags):
        exp = [0] * 7
        if node == 2:
            exp[6] = 1
        if node == 7:
            exp[0] = 1
        assert f == exp, (node, f)


def test_c_oracle_k3_dc(orc):
    ec.check_topology_scenarios(orc, 'default14_for_tests_beta', list(range(1, 15)), 7, _basic_topology_policy)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_c_oracle_default118(orc, solver):
    ec.check_do_nothing(orc, 'default118', solver, steps=5, batch=2)
