// Build shim: the subset of wjakob/filesystem's path class used by util/file.cpp:33-100,
// implemented over std::filesystem.
#pragma once
#include <filesystem>
#include <string>
namespace filesystem {
class path {
  public:
    path() {}
    path(const std::string &s) : p_(s) {}
    path(const char *s) : p_(s) {}
    path(const std::filesystem::path &p) : p_(p) {}
    bool is_absolute() const { return p_.is_absolute(); }
    bool empty() const { return p_.empty(); }
    bool exists() const { return std::filesystem::exists(p_); }
    bool is_directory() const { return std::filesystem::is_directory(p_); }
    bool is_file() const { return std::filesystem::is_regular_file(p_); }
    std::string extension() const {
        std::string e = p_.extension().string();
        return e.empty() ? e : e.substr(1);
    }
    std::string filename() const { return p_.filename().string(); }
    path parent_path() const { return path(p_.parent_path()); }
    path make_absolute() const { return path(std::filesystem::absolute(p_)); }
    path operator/(const path &o) const { return path(p_ / o.p_); }
    std::string str() const { return p_.string(); }
    operator std::string() const { return p_.string(); }
  private:
    std::filesystem::path p_;
};
}  // namespace filesystem
