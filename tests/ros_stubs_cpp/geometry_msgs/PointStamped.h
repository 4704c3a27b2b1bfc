#pragma once
#include <geometry_msgs/Pose.h>
namespace geometry_msgs {
struct PointStamped {
  std_msgs::Header header;
  Point point;
};
}  // namespace geometry_msgs
