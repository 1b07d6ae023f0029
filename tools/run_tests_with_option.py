"""python tools/run_tests_with_option.py <option> <value> <pytest args...>: run tests with a library option preset."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from textflux_amd import ops
ops.set_option(sys.argv[1], int(sys.argv[2]))
sys.exit(pytest.main(sys.argv[3:]))
