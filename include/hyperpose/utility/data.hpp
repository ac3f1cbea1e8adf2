// hyperpose::feature_map_t / internal_t — reference include/hyperpose/utility/data.hpp:17-67.
// A feature_map_t owns a HOST copy (API compatibility with the reference, src/tensorrt.cpp:423-428).  The
// MI355X fast path keeps feature maps in HBM (dnn::hip_engine::inference_device + parser::paf::process_device).
#pragma once
#include <memory>
#include <ostream>
#include <string>
#include <vector>

#include "human.hpp"

namespace hyperpose {

struct feature_map_t {
public:
    feature_map_t(std::string name, std::unique_ptr<char[]>&& tensor, std::vector<int> shape)
        : m_name(std::move(name)), m_data(std::move(tensor)), m_shape(std::move(shape)) {}
    friend std::ostream& operator<<(std::ostream& out, const feature_map_t& map)
    {
        out << map.m_name << ":[";
        for (auto& s : map.m_shape)
            out << s << ", ";
        return out << ']';
    }
    inline const std::string& name() const { return m_name; }
    inline const std::vector<int>& shape() const { return m_shape; }
    template <typename T>
    inline const T* view() const { return reinterpret_cast<T*>(m_data.get()); }

private:
    std::string m_name;
    std::unique_ptr<char[]> m_data;
    std::vector<int> m_shape;
};

using internal_t = std::vector<feature_map_t>;

} // namespace hyperpose
