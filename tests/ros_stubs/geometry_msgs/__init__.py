"""ROS message package stand-in (TEST INFRASTRUCTURE ONLY)."""
