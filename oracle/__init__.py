"""Test-only CPU oracle for the ETPNav planner hot path (never imported by etpnav_amd)."""
