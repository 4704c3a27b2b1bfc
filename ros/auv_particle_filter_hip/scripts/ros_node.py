#!/usr/bin/env python3
"""Entry point of the catkin package: the rospy wrapper lives in smarc_navigation_amd.ros_node."""
from smarc_navigation_amd.ros_node import main

if __name__ == '__main__':
    main()
