#pragma once
#include <array>
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct Point {
  double x = 0, y = 0, z = 0;
};
struct Vector3 {
  double x = 0, y = 0, z = 0;
};
struct Quaternion {
  double x = 0, y = 0, z = 0, w = 0;
};
struct Pose {
  Point position;
  Quaternion orientation;
};
struct PoseWithCovariance {
  Pose pose;
  std::array<double, 36> covariance{};
};
struct Twist {
  Vector3 linear, angular;
};
struct TwistWithCovariance {
  Twist twist;
  std::array<double, 36> covariance{};
};
struct Transform {
  Vector3 translation;
  Quaternion rotation;
};
}  // namespace geometry_msgs
