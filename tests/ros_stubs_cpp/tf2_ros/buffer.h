#pragma once
#include <cmath>
#include <map>
#include <geometry_msgs/PointStamped.h>
#include <geometry_msgs/TransformStamped.h>
#include <tf2/exceptions.h>
namespace tf2_ros {
namespace stub {
// static transforms of the scenario: (target, source) -> transform taking source-frame points into the target frame
inline std::map<std::pair<std::string, std::string>, geometry_msgs::Transform>& transforms() {
  static std::map<std::pair<std::string, std::string>, geometry_msgs::Transform> t;
  return t;
}
inline std::vector<geometry_msgs::TransformStamped>& broadcasts() {
  static std::vector<geometry_msgs::TransformStamped> b;
  return b;
}
}  // namespace stub
class Buffer {
 public:
  geometry_msgs::TransformStamped lookupTransform(const std::string& target, const std::string& source, const ros::Time&,
                                                  const ros::Duration&) const {
    const auto it = stub::transforms().find({target, source});
    if (it == stub::transforms().end()) throw tf2::LookupException("no transform " + target + " <- " + source);
    geometry_msgs::TransformStamped t;
    t.header.frame_id = target;
    t.child_frame_id = source;
    t.transform = it->second;
    return t;
  }
  // (tf2_geometry_msgs' doTransform for a stamped point)
  geometry_msgs::PointStamped& transform(const geometry_msgs::PointStamped& in, geometry_msgs::PointStamped& out,
                                         const std::string& target, ros::Duration = ros::Duration(0.0)) const {
    const geometry_msgs::Transform T = lookupTransform(target, in.header.frame_id, in.header.stamp, ros::Duration(0.0)).transform;
    const double x = T.rotation.x, y = T.rotation.y, z = T.rotation.z, w = T.rotation.w;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    const double p[3] = {in.point.x, in.point.y, in.point.z};
    out.header = in.header;
    out.header.frame_id = target;
    out.point.x = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + T.translation.x;
    out.point.y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + T.translation.y;
    out.point.z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + T.translation.z;
    return out;
  }
};
}  // namespace tf2_ros
