#pragma once
#include <cstring>
#include <stdexcept>
#include <sensor_msgs/PointCloud2.h>
namespace sensor_msgs {
// one field of every point, in storage order (rows of row_step bytes, points of point_step bytes)
template <class T>
class PointCloud2ConstIterator {
 public:
  PointCloud2ConstIterator(const PointCloud2& c, const std::string& field) : c_(&c) {
    bool found = false;
    for (const PointField& f : c.fields)
      if (f.name == field) {
        off_ = f.offset;
        found = true;
      }
    if (!found) throw std::runtime_error("Field " + field + " does not exist");
    n_ = (size_t)c.width * c.height;
  }
  T operator*() const {
    const size_t row = c_->width ? i_ / c_->width : 0, col = c_->width ? i_ % c_->width : 0;
    T v;
    std::memcpy(&v, c_->data.data() + row * c_->row_step + col * c_->point_step + off_, sizeof(T));
    return v;
  }
  PointCloud2ConstIterator& operator++() {
    ++i_;
    return *this;
  }
  bool operator!=(const PointCloud2ConstIterator& o) const { return i_ != o.i_; }
  PointCloud2ConstIterator end() const {
    PointCloud2ConstIterator e(*this);
    e.i_ = n_;
    return e;
  }

 private:
  const PointCloud2* c_;
  size_t off_ = 0, i_ = 0, n_ = 0;
};
}  // namespace sensor_msgs
