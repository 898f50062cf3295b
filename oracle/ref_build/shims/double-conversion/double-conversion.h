// Build shim (test infrastructure only): minimal stand-in for google/double-conversion,
// just the API surface the reference uses at parser.cpp:126-128,408-410 and
// util/print.cpp:15-42. Backed by strtof/strtod/snprintf (all correctly rounded).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
namespace double_conversion {
class StringToDoubleConverter {
  public:
    enum Flags { NO_FLAGS = 0, ALLOW_HEX = 1 };
    StringToDoubleConverter(int, double empty, double junk, const char *, const char *)
        : empty_(empty), junk_(junk) {}
    double StringToDouble(const char *buf, int len, int *processed) const {
        std::string s(buf, len);
        char *end;
        double v = strtod(s.c_str(), &end);
        *processed = int(end - s.c_str());
        return *processed ? v : junk_;
    }
    float StringToFloat(const char *buf, int len, int *processed) const {
        std::string s(buf, len);
        char *end;
        float v = strtof(s.c_str(), &end);
        *processed = int(end - s.c_str());
        return *processed ? v : float(junk_);
    }
  private:
    double empty_, junk_;
};
class StringBuilder {
  public:
    StringBuilder(char *buf, int size) : buf_(buf), size_(size), pos_(0) {}
    int position() const { return pos_; }
    void Add(const char *s) {
        int n = (int)strlen(s);
        if (pos_ + n < size_) { memcpy(buf_ + pos_, s, n); pos_ += n; buf_[pos_] = 0; }
    }
  private:
    char *buf_; int size_, pos_;
};
class DoubleToStringConverter {
  public:
    enum Flags { NO_FLAGS = 0 };
    DoubleToStringConverter(int, const char *inf, const char *nan, char, int, int, int, int)
        : inf_(inf), nan_(nan) {}
    bool ToShortest(double v, StringBuilder *sb) const { return emit<double>(v, 17, sb); }
    bool ToShortestSingle(float v, StringBuilder *sb) const { return emit<float>(v, 9, sb); }
  private:
    template <typename T>
    bool emit(T v, int maxp, StringBuilder *sb) const {
        if (v != v) { sb->Add(nan_); return true; }
        if (v - v != 0) { if (v < 0) sb->Add("-"); sb->Add(inf_); return true; }
        char tmp[64];
        for (int p = 1; p <= maxp; ++p) {
            snprintf(tmp, sizeof(tmp), "%.*g", p, (double)v);
            T back = sizeof(T) == 4 ? (T)strtof(tmp, nullptr) : (T)strtod(tmp, nullptr);
            if (back == v) break;
        }
        sb->Add(tmp);
        return true;
    }
    const char *inf_, *nan_;
};
}  // namespace double_conversion
