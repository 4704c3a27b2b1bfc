#pragma once
#include <tf2_ros/buffer.h>
