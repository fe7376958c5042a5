// flags.hpp -- tiny argv parser accepting the reference demos' gflags syntax: --name=value, --name value,
// -name=value, --flagfile=<path> (one flag per line, '#' comments), boolean --name / --noname.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

class Flags {
public:
    void Parse(int argc, char** argv) {
        std::vector<std::string> args(argv + 1, argv + argc);
        ParseList(args);
    }
    double Double(const std::string& name, double def) const {
        auto it = kv_.find(name);
        return it == kv_.end() ? def : std::atof(it->second.c_str());
    }
    long Int(const std::string& name, long def) const {
        auto it = kv_.find(name);
        return it == kv_.end() ? def : std::atol(it->second.c_str());
    }
    // gflags booleans: --name, --name=true|false|1|0, --noname
    bool Bool(const std::string& name, bool def) const {
        auto it = kv_.find(name);
        if (it != kv_.end()) return !(it->second == "false" || it->second == "0" || it->second == "no");
        if (kv_.find("no" + name) != kv_.end()) return false;
        return def;
    }
    std::string String(const std::string& name, const std::string& def) const {
        auto it = kv_.find(name);
        return it == kv_.end() ? def : it->second;
    }
private:
    void ParseList(const std::vector<std::string>& args) {
        for (size_t i = 0; i < args.size(); ++i) {
            std::string a = args[i];
            if (a.empty() || a[0] == '#') continue;
            if (a.rfind("--", 0) == 0) a = a.substr(2);
            else if (a.rfind("-", 0) == 0) a = a.substr(1);
            else continue;
            std::string name = a, value = "true";
            size_t eq = a.find('=');
            if (eq != std::string::npos) { name = a.substr(0, eq); value = a.substr(eq + 1); }
            else if (i + 1 < args.size() && args[i + 1].rfind("-", 0) != 0) value = args[++i];
            if (name == "flagfile") {
                std::ifstream f(value);
                if (!f) throw std::runtime_error("can't open flagfile " + value);
                std::vector<std::string> lines;
                for (std::string line; std::getline(f, line);) {
                    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
                    if (!line.empty()) lines.push_back(line);
                }
                ParseList(lines);
            } else {
                kv_[name] = value;
            }
        }
    }
    std::map<std::string, std::string> kv_;
};
