"""Scene-plugin registry: a scene file subclasses `BuildFunctionBase`, names itself with
`BUILD_FUNCTION_NAME` and provides `build_function(cls, world, viewmode)`
(/root/reference/buildfunctions/buildfunctionbase.py:1-8)."""


class BuildFunctionBase(object):
    BUILD_FUNCTION_NAME = None

    @classmethod
    def get_build_function(cls, name):
        # the most recently defined scene of a name wins, so re-importing a scene file
        # (or shadowing a bundled scene with a user's) behaves predictably
        found = None
        for scene in BuildFunctionBase.__subclasses__():
            if scene.BUILD_FUNCTION_NAME == name:
                found = scene.build_function
        return found
